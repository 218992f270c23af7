// FM_GPU.java -- drop-in for carskit.alg.cars.adaptation.dependent.FM, registered next to "fm" (src/carskit/main/CARSKit.java:742)
// as "fm_gpu".  The reference keeps w0 / w / V in PRIVATE fields (FM.java:40-47), so this class cannot extend FM: it extends
// ContextRecommender like FM does, owns the same three containers, initialises them with the same calls in the same order
// (FM.java:57-74, minus the size x k cache `Q`, which only buildModel() uses), trains with cmi_fm_train (the pre-pass + numIters
// ALS sweeps of FM.java:115-220) and predicts with the model equation in its pairwise form over the <= 3 non-zero features
// of a rating (FM.java:76-113 builds the dense p-vector and loops over it: same value, O(k) instead of O(p k)).
// No JDK in this image: not compiled by javac here.  EXECUTED under the Java-source interpreter (oracle/check_java_binding.py,
// tests/test_java_binding_exec.py: buildModel() bit-identical to FM.buildModel(), predict() within 1e-12 of FM.predict()).
package carskit.alg.gpu;

import carskit.data.structure.SparseMatrix;
import carskit.generic.ContextRecommender;
import librec.data.DenseMatrix;
import librec.data.DenseVector;
import java.util.Map;

public class FM_GPU extends ContextRecommender {
    private double w0;
    private int p, k;
    private DenseVector w;   // size p = numUsers + numItems + numConditions
    private DenseMatrix V;   // p x k
    private final float regLw, regLf;

    public FM_GPU(SparseMatrix trainMatrix, SparseMatrix testMatrix, int fold) {
        super(trainMatrix, testMatrix, fold);
        this.algoName = "FM_GPU";
        regLw = algoOptions.getFloat("-lw");
        regLf = algoOptions.getFloat("-lf");
    }

    @Override
    protected void initModel() throws Exception {
        super.initModel();
        k = numFactors;
        p = numUsers + numItems + numConditions;
        w0 = 0;
        w = new DenseVector(p);
        w.init();                        // uniform(0,1), then ...
        V = new DenseMatrix(p, k);
        V.init(initMean, initStd);       // ... gaussian: the reference's draw order from librec's static RNG
    }

    @Override
    protected void buildModel() throws Exception {
        int numDims = rateDao.numContextDims();
        long h = NativeMF.fmCreate(k, numUsers, numItems, numConditions, numDims, GpuSupport.deviceFor(fold), 0);
        try {
            int[][] ui = GpuSupport.pairMaps(rateDao, trainMatrix.numRows());
            NativeMF.fmSetHparams(h, regLw, regLf, trainMatrix.size());
            NativeMF.fmSetRatingsCsr(h, trainMatrix.getRowPointers(), trainMatrix.getColumnIndices(), trainMatrix.getData(), ui[0], ui[1]);
            NativeMF.fmSetModel(h, w0, w.getData(), Rows.of(V));
            NativeMF.fmTrain(h, numIters);                       // replaces FM.java:115-220 (no early stop there either)
            w0 = NativeMF.fmGetModel(h, w.getData(), Rows.of(V));
        } finally {
            NativeMF.fmDestroy(h);
        }
    }

    @Override
    protected Map<Measure, Double> evalRankings() throws Exception {
        if (!GpuSupport.rankOnGpu() || isDiverseUsed) return super.evalRankings();   // the reference's loop (Recommender.java:672-955)
        long h = NativeMF.fmCreate(k, numUsers, numItems, numConditions, rateDao.numContextDims(), GpuSupport.deviceFor(fold), 0);
        try {                                                                         // -Dcarskit.gpu.rank=true: cmi_fm_eval_rankings
            int[][] ui = GpuSupport.pairMaps(rateDao, trainMatrix.numRows());
            NativeMF.fmSetHparams(h, regLw, regLf, trainMatrix.size());
            NativeMF.fmSetRatingsCsr(h, trainMatrix.getRowPointers(), trainMatrix.getColumnIndices(), trainMatrix.getData(), ui[0], ui[1]);
            NativeMF.fmSetModel(h, w0, w.getData(), Rows.of(V));
            Object[] tr = GpuSupport.tuples(trainMatrix, rateDao), te = GpuSupport.tuples(testMatrix, rateDao);
            double[] out = NativeMF.fmEvalRankings(h, (int[]) tr[0], (int[]) tr[1], (int[]) tr[2], (double[]) tr[3],
                                                   (int[]) te[0], (int[]) te[1], (int[]) te[2], (double[]) te[3], binThold, numRecs, numIgnore,
                                                   evalStrategy.equals("uc") ? NativeMF.RANK_UC : NativeMF.RANK_UCU);
            return GpuSupport.rankingMeasures(out);
        } finally {
            NativeMF.fmDestroy(h);
        }
    }

    @Override
    protected double predict(int u, int j, int c) throws Exception {
        // features: user u (1), item numUsers + j (1), and -- the reference's index quirk, FM.java:81-86 -- the context
        // COMBINATION id c as feature numUsers + numItems + c with value 1 / numContextDims, only if c < numConditions
        int iu = u, ij = numUsers + j, ic = numUsers + numItems + c;
        boolean hasC = c >= 0 && c < numConditions;
        double xc = hasC ? 1.0 / rateDao.numContextDims() : 0.0;
        double pred = w0 + w.get(iu) + w.get(ij) + (hasC ? w.get(ic) * xc : 0.0);
        for (int f = 0; f < k; f++) {
            double a = V.get(iu, f), b = V.get(ij, f), d = hasC ? V.get(ic, f) * xc : 0.0;
            double s = a + b + d;
            pred += 0.5 * (s * s - a * a - b * b - d * d);
        }
        return pred;
    }
}

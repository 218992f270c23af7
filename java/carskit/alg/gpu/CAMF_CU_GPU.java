// CAMF_CU_GPU.java -- drop-in for carskit.alg.cars.adaptation.dependent.dev.CAMF_CU: identical constructor, initModel() and predict()
// (inherited); buildModel() runs on the GPU through GpuSupport.buildModel, evalRatings() reads the device model while training
// is in progress (so `--early-stop MAE|RMSE` sees the live model, IterativeRecommender.java:156-161).  Registered in the reference's
// factory switch next to "camf_cu" (src/carskit/main/CARSKit.java:704) as "camf_cu_gpu".
// No JDK in this image: not compiled by javac here.  EXECUTED under the Java-source interpreter over the reference's own class chain
// (oracle/check_java_binding.py, tests/test_java_binding_exec.py: bit-identical to the reference's buildModel()); the NativeMF calls are
// also checked as text (tests/test_java_binding_text.py).
package carskit.alg.gpu;

import carskit.alg.cars.adaptation.dependent.dev.CAMF_CU;
import carskit.data.structure.SparseMatrix;
import java.util.List;
import java.util.Map;

public class CAMF_CU_GPU extends CAMF_CU implements GpuHost {
    private long gpuHandle = 0L;

    public CAMF_CU_GPU(SparseMatrix trainMatrix, SparseMatrix testMatrix, int fold) {
        super(trainMatrix, testMatrix, fold);
        this.algoName = "CAMF_CU_GPU";
    }

    @Override
    protected void buildModel() throws Exception {
        GpuSupport.buildModel(this);   // replaces CAMF_CU.java:76-120
    }

    @Override
    protected Map<Measure, Double> evalRatings() throws Exception {
        if (gpuHandle == 0L) return super.evalRatings();                       // after buildModel(): the copied-back model
        return GpuSupport.evalResident(gpuHandle, minRate, maxRate);           // during buildModel(): the model on the device
    }

    @Override
    protected Map<Measure, Double> evalRankings() throws Exception {
        if (!GpuSupport.rankOnGpu() || isDiverseUsed) return super.evalRankings();   // the reference's loop (Recommender.java:672-955)
        return GpuSupport.evalRankings(this, binThold, numRecs, numIgnore, evalStrategy);   // -Dcarskit.gpu.rank=true: cmi_eval_rankings
    }

    // ---- GpuHost: the protected members of the reference classes GpuSupport needs ----
    public int modelId() { return NativeMF.CAMF_CU; }
    public int createFlags() { return 0; }
    public int factors() { return numFactors; }
    public int users() { return numUsers; }
    public int items() { return numItems; }
    public int conditions() { return numConditions; }
    public int foldId() { return fold; }
    public SparseMatrix contextualTrain() { return trainMatrix; }
    public SparseMatrix contextualTest() { return testMatrix; }
    public librec.data.SparseMatrix train2D() { return train; }
    public List<Integer> conditionsOf(int ctx) { return getConditions(ctx); }
    public double[] regularizers() { return new double[] {regU, regI, regB, regC}; }
    public double mean() { return globalMean; }
    public double minRating() { return minRate; }
    public double maxRating() { return maxRate; }
    public int iterations() { return numIters; }
    public double learnRate() { return lRate; }
    public boolean evaluatesDuringTraining() { return earlyStopMeasure != null && earlyStopMeasure != Measure.Loss; }
    public void handle(long h) { gpuHandle = h; }
    public boolean epochDone(int iter, double epochLoss) throws Exception {
        loss = epochLoss;
        return isConverged(iter);      // unchanged reference code: IterativeRecommender.java:145-199
    }
    public void prepare(long h) {}
    public void copyIn(long h) {
        Dev.setMatrix(h, NativeMF.P, Rows.of(P));
        Dev.setMatrix(h, NativeMF.Q, Rows.of(Q));
        Dev.setVector(h, NativeMF.ITEM_BIAS, itemBias.getData());
        Dev.setMatrix(h, NativeMF.UC_BIAS, Rows.of(ucBias));
    }
    public void copyOut(long h) {
        Dev.getMatrix(h, NativeMF.P, Rows.of(P));
        Dev.getMatrix(h, NativeMF.Q, Rows.of(Q));
        Dev.getVector(h, NativeMF.ITEM_BIAS, itemBias.getData());
        Dev.getMatrix(h, NativeMF.UC_BIAS, Rows.of(ucBias));
    }
}

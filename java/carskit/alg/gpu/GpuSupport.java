// GpuSupport.java -- the part of buildModel()/evalRatings() every *_GPU recommender shares.  Only marshalling and the
// reference's own control flow (the epoch loop with isConverged()); all arithmetic is behind NativeMF.
// No JDK in the build image: not compiled by javac here; executed under the Java-source interpreter (tests/test_java_binding_exec.py).
package carskit.alg.gpu;

import carskit.data.processor.DataDAO;
import carskit.data.structure.SparseMatrix;
import carskit.generic.Recommender;
import carskit.generic.Recommender.Measure;
import java.util.ArrayList;
import java.util.HashMap;
import java.util.List;
import java.util.Map;

final class GpuSupport {
    private GpuSupport() {}

    /** fold -> GPU round robin (the reference runs one thread per fold, CARSKit.java:395-412); -Dcarskit.gpus=N */
    static int deviceFor(int fold) {
        return Math.max(0, fold - 1) % Math.max(1, Integer.getInteger("carskit.gpus", 1));
    }

    /** {uiUser, uiItem}: DataDAO.getUserIdFromUI / getItemIdFromUI (DataDAO.java:1038-1046) for every pair id */
    static int[][] pairMaps(DataDAO dao, int numPairs) {
        int[] uiUser = new int[numPairs], uiItem = new int[numPairs];
        for (int ui = 0; ui < numPairs; ui++) {
            uiUser[ui] = dao.getUserIdFromUI(ui);
            uiItem[ui] = dao.getItemIdFromUI(ui);
        }
        return new int[][] {uiUser, uiItem};
    }

    /** {ctxPtr, ctxConds}: getConditions(ctx) for every context id, flattened once (ContextRecommender.java:53-61) */
    static int[][] contextTable(GpuHost r, int numCtx) {
        int[] ctxPtr = new int[numCtx + 1];
        ArrayList<Integer> conds = new ArrayList<Integer>();
        for (int c = 0; c < numCtx; c++) {
            conds.addAll(r.conditionsOf(c));
            ctxPtr[c + 1] = conds.size();
        }
        int[] ctxConds = new int[conds.size()];
        for (int i = 0; i < ctxConds.length; i++) ctxConds[i] = conds.get(i);
        return new int[][] {ctxPtr, ctxConds};
    }

    /** the tuples `for (MatrixEntry me : m)` yields, as {u, j, ctx} + rates (test side: setEvalRatings / evalRankings) */
    static Object[] tuples(SparseMatrix m, DataDAO dao) {
        int[] rowPtr = m.getRowPointers(), colInd = m.getColumnIndices();
        double[] data = m.getData();
        int n = colInd.length;
        int[] u = new int[n], j = new int[n], ctx = new int[n];
        double[] r = new double[n];
        for (int row = 0; row + 1 < rowPtr.length; row++)
            for (int q = rowPtr[row]; q < rowPtr[row + 1]; q++) {
                u[q] = dao.getUserIdFromUI(row);
                j[q] = dao.getItemIdFromUI(row);
                ctx[q] = colInd[q];
                r[q] = data[q];
            }
        return new Object[] {u, j, ctx, r};
    }

    /** -Dcarskit.shards=N: ONE recommender trained over N GPUs (ratings cut by user inside the library, item-side containers merged
     *  over RCCL after every epoch: cmi_group_*).  Default 1 = the single-GPU order-exact path.  Not the same thing as
     *  -Dcarskit.gpus (folds -> GPUs, the reference's own parallelism, CARSKit.java:395-412). */
    static int shards() { return Math.max(1, Integer.getInteger("carskit.shards", 1)); }

    /** The replacement of the reference's buildModel(): upload once, run the epoch loop with the UNCHANGED Java isConverged()
     *  (bold driver, decay, early stop, NaN exit: IterativeRecommender.java:145-229), copy the model back. */
    static void buildModel(GpuHost r) throws Exception {
        boolean serialChain = (r.createFlags() & NativeMF.FLAG_SCHED_SERIAL) != 0;   // CAMF_C, SVD++, CAMF_ICS/LCS/MCS: not sharded
        if (shards() > 1 && !serialChain) {
            buildModelSharded(r, shards());
            return;
        }
        boolean twoD = r.modelId() == NativeMF.BIASEDMF || r.modelId() == NativeMF.PMF || r.modelId() == NativeMF.SVDPP;
        long h = NativeMF.create(r.modelId(), r.factors(), r.users(), r.items(), r.conditions(), deviceFor(r.foldId()), r.createFlags());
        try {
            DataDAO dao = Recommender.rateDao;
            upload(r, h, twoD);
            if (r.evaluatesDuringTraining() && r.contextualTest() != null) {
                // `--early-stop MAE|RMSE`: isConverged() scores the test set after EVERY epoch; the Java-side containers are
                // stale until copyOut, so evalRatings() of the drop-in reads the device model instead (evalResident below)
                Object[] t = tuples(r.contextualTest(), dao);
                NativeMF.setEvalRatings(h, (int[]) t[0], (int[]) t[1], twoD ? null : (int[]) t[2], (double[]) t[3]);
            }
            r.handle(h);
            for (int iter = 1; iter <= r.iterations(); iter++) {
                double loss = NativeMF.trainEpoch(h, r.learnRate());   // replaces e.g. CAMF_CI.java:79-123
                if (r.epochDone(iter, loss)) break;                    // unchanged Java: loss = ...; isConverged(iter)
            }
            r.copyOut(h);   // predict() / evalRatings() / evalRankings() / saveModel() keep working unchanged afterwards
        } finally {
            r.handle(0L);
            NativeMF.destroy(h);
        }
    }

    /** everything a native call needs before it can train or score: sim params, the rating matrix (and with it the context table),
     *  hyper-parameters, the model containers */
    static void upload(GpuHost r, long h, boolean twoD) {
        DataDAO dao = Recommender.rateDao;
        SparseMatrix tm = r.contextualTrain();
        // `cv -p on` (CARSKit.java:395-412): -Dcarskit.folds.per.gpu=F tells the library how many fold threads train side by side on this
        // GPU (cmi_set_device_share: their persistent owner epochs then run concurrently instead of taking turns); default 1
        int share = Math.max(1, Integer.getInteger("carskit.folds.per.gpu", 1));
        if (share > 1) NativeMF.setDeviceShare(h, share);
        r.prepare(h);
        if (twoD) {
            librec.data.SparseMatrix t2 = r.train2D();
            NativeMF.setRatings2D(h, t2.getRowPointers(), t2.getColumnIndices(), t2.getData());
        } else {
            int[][] ui = pairMaps(dao, tm.numRows());
            int[][] ct = contextTable(r, tm.numColumns());
            NativeMF.setRatingsCsr(h, tm.getRowPointers(), tm.getColumnIndices(), tm.getData(), ui[0], ui[1], ct[0], ct[1]);
        }
        double[] reg = r.regularizers();
        NativeMF.setHparams(h, reg[0], reg[1], reg[2], reg[3], r.mean());
        r.copyIn(h);
    }

    /** -Dcarskit.gpu.rank=true: evalRankings() of the drop-ins scores on the GPU (cmi_eval_rankings) instead of walking
     *  candidates x queries in Java (Recommender.java:672-955); off by default, `-diverse` stays on the Java path. */
    static boolean rankOnGpu() { return Boolean.getBoolean("carskit.gpu.rank"); }

    /** evalRankings() on the native side: the model as it stands in the Java object is uploaded to a fresh handle, the train and test
     *  matrices go over as (u, j, ctx, rate) tuples, the 21 measures come back in Recommender.Measure order. */
    static Map<Measure, Double> evalRankings(GpuHost r, double binThold, int numRecs, int numIgnore, String evalStrategy) throws Exception {
        boolean twoD = r.modelId() == NativeMF.BIASEDMF || r.modelId() == NativeMF.PMF || r.modelId() == NativeMF.SVDPP;
        long h = NativeMF.create(r.modelId(), r.factors(), r.users(), r.items(), r.conditions(), deviceFor(r.foldId()), r.createFlags());
        try {
            upload(r, h, twoD);
            DataDAO dao = Recommender.rateDao;
            Object[] tr = tuples(r.contextualTrain(), dao), te = tuples(r.contextualTest(), dao);
            double[] out = NativeMF.evalRankings(h, (int[]) tr[0], (int[]) tr[1], (int[]) tr[2], (double[]) tr[3],
                                                 (int[]) te[0], (int[]) te[1], (int[]) te[2], (double[]) te[3], binThold, numRecs, numIgnore,
                                                 evalStrategy.equals("uc") ? NativeMF.RANK_UC : NativeMF.RANK_UCU);
            return rankingMeasures(out);
        } finally {
            NativeMF.destroy(h);
        }
    }

    /** The same flow over a cmi_group: the library shards the CSR arrays by user, every epoch returns the GLOBAL loss, so the
     *  unchanged isConverged() steers all shards.  `--early-stop MAE|RMSE`: the test tuples are routed once to the shards that own
     *  their users and stay on the devices (cmi_group_set_eval_ratings); isConverged()'s evalRatings() reads the shards' sums
     *  (cmi_group_eval_resident) through the drop-in's live handle, exactly as on one GPU. */
    static void buildModelSharded(GpuHost r, int nShards) throws Exception {
        boolean twoD = r.modelId() == NativeMF.BIASEDMF || r.modelId() == NativeMF.PMF;
        long g = NativeMF.groupCreate(r.modelId(), r.factors(), r.users(), r.items(), r.conditions(), nShards, null, r.createFlags());
        try {
            DataDAO dao = Recommender.rateDao;
            SparseMatrix tm = r.contextualTrain();
            double[] reg = r.regularizers();
            NativeMF.groupSetHparams(g, reg[0], reg[1], reg[2], reg[3], r.mean());
            if (twoD) {
                librec.data.SparseMatrix t2 = r.train2D();
                NativeMF.groupSetRatings2D(g, t2.getRowPointers(), t2.getColumnIndices(), t2.getData());
            } else {
                int[][] ui = pairMaps(dao, tm.numRows());
                int[][] ct = contextTable(r, tm.numColumns());
                NativeMF.groupSetRatingsCsr(g, tm.getRowPointers(), tm.getColumnIndices(), tm.getData(), ui[0], ui[1], ct[0], ct[1]);
            }
            r.copyIn(Dev.ofGroup(g));
            // the mean merge slows convergence per epoch (1.2x / 1.4x / 1.6x at 2 / 4 / 8 shards); sqrt(N) on the LOCAL rate recovers
            // it to <= 1.2x while isConverged()'s bold driver keeps steering the base rate (DESIGN.md section 7); -Dcarskit.shards.lrscale=1 opts out
            NativeMF.groupSetLrScale(g, Double.parseDouble(System.getProperty("carskit.shards.lrscale", Double.toString(Math.sqrt(nShards)))));
            if (r.evaluatesDuringTraining() && r.contextualTest() != null) {
                Object[] t = tuples(r.contextualTest(), dao);
                NativeMF.groupSetEvalRatings(g, (int[]) t[0], (int[]) t[1], twoD ? null : (int[]) t[2], (double[]) t[3]);
                r.handle(-g);   // a drop-in's live handle < 0 = a cmi_group (native handles are user-space addresses: positive)
            }
            for (int iter = 1; iter <= r.iterations(); iter++) {
                double loss = NativeMF.groupTrainEpoch(g, r.learnRate());
                if (r.epochDone(iter, loss)) break;
            }
            r.copyOut(Dev.ofGroup(g));
        } finally {
            r.handle(0L);
            NativeMF.groupDestroy(g);
        }
    }

    /** evalRatings() while the native handle is live: the same measures the reference computes (Recommender.java:504-594),
     *  from the model on the device. */
    static Map<Measure, Double> evalResident(long h, double minRate, double maxRate) {
        double[] m = h < 0 ? NativeMF.groupEvalResident(-h, minRate, maxRate) : NativeMF.evalResident(h, minRate, maxRate);
        Map<Measure, Double> out = new HashMap<Measure, Double>();
        out.put(Measure.MAE, m[0]);
        out.put(Measure.RMSE, m[1]);
        out.put(Measure.NMAE, m[2]);
        out.put(Measure.rMAE, m[3]);
        out.put(Measure.rRMSE, m[4]);
        out.put(Measure.MPE, 0.0);
        return out;
    }

    /** cmi_eval_rankings out[21] -> the reference's measure map (Recommender.java:930-960) */
    static Map<Measure, Double> rankingMeasures(double[] out) {
        Measure[] order = {Measure.Pre5, Measure.Pre10, Measure.PreN, Measure.Rec5, Measure.Rec10, Measure.RecN, Measure.AUC5,
                           Measure.AUC10, Measure.AUCN, Measure.MAP5, Measure.MAP10, Measure.MAPN, Measure.NDCG5, Measure.NDCG10,
                           Measure.NDCGN, Measure.MRR5, Measure.MRR10, Measure.MRRN, Measure.D5, Measure.D10, Measure.DN};
        Map<Measure, Double> m = new HashMap<Measure, Double>();
        for (int i = 0; i < order.length; i++) m.put(order[i], out[i]);
        return m;
    }

    static List<Integer> none() { return new ArrayList<Integer>(); }
}

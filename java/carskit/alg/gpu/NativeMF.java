// NativeMF.java -- Java side of the JNI binding to libcarskit_mi355x.so (include/carskit_mi355x.h).
// The build image and the GPU box have no JDK, so this file is not compiled by javac here (see INTEGRATION.md; the classes that call it
// are executed under the Java-source interpreter with these natives bound to a stand-in: tests/test_java_binding_exec.py);
// tests/test_java_binding_text.py checks, as text, that every native below has exactly one
// Java_carskit_alg_gpu_NativeMF_<name> definition in jni/carskit_jni.cpp with the matching JNI type signature.
// One static native method per C-ABI entry point a Java host needs, same argument meaning; no logic.
package carskit.alg.gpu;

public final class NativeMF {
    static { System.loadLibrary("carskit_mi355x_jni"); }

    private NativeMF() {}

    public static final int BIASEDMF = 0, CAMF_C = 1, CAMF_CI = 2, CAMF_CU = 3, CAMF_CUCI = 4, PMF = 5;
    /** the remaining SGD recommenders of the family (CARSKit.java:469,708-712): serial chains, FLAG_SCHED_SERIAL required */
    public static final int SVDPP = 6, CAMF_ICS = 7, CAMF_LCS = 8, CAMF_MCS = 9;
    public static final int P = 0, Q = 1, USER_BIAS = 2, ITEM_BIAS = 3, COND_BIAS = 4, UC_BIAS = 5, IC_BIAS = 6;
    /** SVD++ Y (numItems x k); ccMatrix_ICS as the full symmetric matrix; cfMatrix_LCS (numConditions x numF); cVector_MCS */
    public static final int Y = 7, CC_MATRIX = 8, CF_MATRIX = 9, C_VECTOR = 10;
    public static final int FLAG_STATE_F64 = 1, FLAG_SCHED_SERIAL = 2, FLAG_STRICT = 4, FLAG_NO_GRAPH = 16;
    /** schedule overrides (include/carskit_mi355x.h); the library picks hub-chain levels / the owner epoch / plain levels by itself */
    public static final int FLAG_SCHED_CHAIN = 0x80, FLAG_NO_CHAIN = 0x100, FLAG_SCHED_OWNER = 0x200, FLAG_NO_OWNER = 0x400;
    /** spoke arena of the hub-chain schedule (picked automatically for spoke tables of 2 GiB and more) */
    public static final int FLAG_SPOKE_ARENA = 0x800, FLAG_NO_ARENA = 0x1000;
    public static final int RANK_UCU = 0, RANK_UC = 1;

    /** cmi_create; returns the handle, throws RuntimeException(cmi_last_error) on failure. */
    public static native long create(int model, int k, int nUsers, int nItems, int nConds, int device, int flags);
    public static native void destroy(long h);
    /** cmi_set_ratings for the contextual models: rowPtr/colInd/data are the live CSR arrays of the librec SparseMatrix
     *  (getRowPointers/getColumnIndices/getData, rows = user-item pair ids, columns = context ids); uiUser/uiItem map a
     *  pair id to its user / item; ctxPtr/ctxConds is getConditions(ctx) for every context id as CSR. */
    public static native void setRatingsCsr(long h, int[] rowPtr, int[] colInd, double[] data, int[] uiUser,
                                            int[] uiItem, int[] ctxPtr, int[] ctxConds);
    /** cmi_set_ratings for BiasedMF / PMF: the 2-D `train` matrix (rows = users, colInd = items; BiasedMF.java:62-66). */
    public static native void setRatings2D(long h, int[] rowPtr, int[] colInd, double[] data);
    /** cmi_set_state / cmi_get_state for a DenseMatrix (double[][] rows flattened by the shim) or DenseVector. */
    public static native void setMatrix(long h, int which, double[][] rows);
    public static native void getMatrix(long h, int which, double[][] rows);
    public static native void setVector(long h, int which, double[] v);
    public static native void getVector(long h, int which, double[] v);
    public static native void setHparams(long h, double regU, double regI, double regB, double regC, double globalMean);
    /** cmi_set_device_share: recommenders training concurrently on this handle's GPU (`cv -p on`); before setRatings*. */
    public static native void setDeviceShare(long h, int instances);
    /** cmi_train_epoch: one pass of the for(MatrixEntry me : trainMatrix) body; returns loss (already *0.5). */
    public static native double trainEpoch(long h, double lRate);
    /** cmi_train: the whole buildModel() loop on the native side (isConverged/updateLRate included); returns the epochs run,
     *  fills losses / lrates (numIters each, may be null). */
    public static native int train(long h, int numIters, double initLRate, double maxLRate, int boldDriver, double decay,
                                   int earlyStop, double[] losses, double[] lrates);
    /** cmi_eval_ratings: {MAE, RMSE, NMAE, rMAE, rRMSE, count}. */
    public static native double[] evalRatings(long h, int[] u, int[] j, int[] ctx, double[] r, double minRate, double maxRate);
    /** cmi_set_eval_ratings + cmi_eval_resident: the test tuples stay on the device for the per-epoch evaluation of
     *  `--early-stop MAE|RMSE` (IterativeRecommender.java:156-161); same six numbers as evalRatings. */
    public static native void setEvalRatings(long h, int[] u, int[] j, int[] ctx, double[] r);
    public static native double[] evalResident(long h, double minRate, double maxRate);
    /** cmi_predict_batch: predict(u, j, c, bound) for n tuples (bound != 0 clamps to [lo, hi]). */
    public static native double[] predictBatch(long h, int[] u, int[] j, int[] ctx, int bound, double lo, double hi);
    /** cmi_eval_rankings: the 21 measures of Recommender.evalRankings (Pre5 .. DN), train/test tuples as (u, j, ctx, rate). */
    public static native double[] evalRankings(long h, int[] tu, int[] tj, int[] tctx, double[] tr, int[] su, int[] sj,
                                               int[] sctx, double[] sr, double binThold, int numRecs, int numIgnore, int strategy);
    /** cmi_save_model / cmi_load_model: all seven containers in one versioned file (IterativeRecommender.java:249-292 forgets
     *  the context tables). */
    public static native void saveModel(long h, String path, double lRate, double lastLoss, int epochsDone);
    /** returns {lRate, lastLoss, epochsDone} as stored with the model */
    public static native double[] loadModel(long h, String path);

    /** cmi_set_sim_params: EmptyContextConditions (ContextRecommender.java:43), `-f` of CAMF_LCS (CAMF_LCS.java:37) and
     *  rateDao.numContextDims() (CAMF_MCS.java:44); call before setRatingsCsr for CAMF_ICS / LCS / MCS. */
    public static native void setSimParams(long h, int numF, int nCtxDims, int[] emptyConds);

    // ---- one recommender sharded by user over several GPUs (cmi_group_*; -Dcarskit.shards=N) ---------------------------
    /** cmi_group_create: devices = one index per shard, or null for round robin over the visible GPUs. */
    public static native long groupCreate(int model, int k, int nUsers, int nItems, int nConds, int nShards, int[] devices, int flags);
    public static native void groupDestroy(long g);
    public static native void groupSetHparams(long g, double regU, double regI, double regB, double regC, double globalMean);
    /** the WHOLE training matrix, as setRatingsCsr / setRatings2D; the library cuts it by user */
    public static native void groupSetRatingsCsr(long g, int[] rowPtr, int[] colInd, double[] data, int[] uiUser, int[] uiItem,
                                                 int[] ctxPtr, int[] ctxConds);
    public static native void groupSetRatings2D(long g, int[] rowPtr, int[] colInd, double[] data);
    /** whole containers: user-side ones are scattered to / gathered from the owning shards, item-side ones replicated */
    public static native void groupSetMatrix(long g, int which, double[][] rows);
    public static native void groupGetMatrix(long g, int which, double[][] rows);
    public static native void groupSetVector(long g, int which, double[] v);
    public static native void groupGetVector(long g, int which, double[] v);
    /** cmi_group_set_lr_scale: local learning rate = lRate x scale (GpuSupport sets sqrt(nShards): near-sequential epochs-to-RMSE) */
    public static native void groupSetLrScale(long g, double scale);
    /** cmi_group_train_epoch: every shard's local pass + the merge of the item-side moves; returns the GLOBAL loss */
    public static native double groupTrainEpoch(long g, double lRate);
    /** cmi_group_eval_ratings: {MAE, RMSE, NMAE, rMAE, rRMSE, count}, test tuples routed to the shard that owns their user */
    public static native double[] groupEvalRatings(long g, int[] u, int[] j, int[] ctx, double[] r, double minRate, double maxRate);
    /** cmi_group_set_eval_ratings + cmi_group_eval_resident: `--early-stop MAE|RMSE` with -Dcarskit.shards=N -- the test tuples go to
     *  the shards that own their users once and stay on the devices; {MAE, RMSE, NMAE, rMAE, rRMSE, count} */
    public static native void groupSetEvalRatings(long g, int[] u, int[] j, int[] ctx, double[] r);
    public static native double[] groupEvalResident(long g, double minRate, double maxRate);

    // ---- FM (src/carskit/alg/cars/adaptation/dependent/FM.java) --------------------------------------------------
    public static native long fmCreate(int k, int nUsers, int nItems, int nConds, int nCtxDims, int device, int flags);
    public static native void fmDestroy(long h);
    public static native void fmSetHparams(long h, double regLw, double regLf, long globalSize);
    /** cmi_fm_set_ratings: the contextual CSR expanded like setRatingsCsr (ctx = context-combination id). */
    public static native void fmSetRatingsCsr(long h, int[] rowPtr, int[] colInd, double[] data, int[] uiUser, int[] uiItem);
    public static native void fmSetModel(long h, double w0, double[] w, double[][] vRows);
    /** cmi_fm_get_model: returns w0, fills w and the rows of V. */
    public static native double fmGetModel(long h, double[] w, double[][] vRows);
    /** cmi_fm_train: the pre-pass + numIters ALS sweeps (FM.java:115-220). */
    public static native void fmTrain(long h, int numIters);
    public static native double[] fmPredictBatch(long h, int[] u, int[] j, int[] ctx, int bound, double lo, double hi);
    public static native double[] fmEvalRankings(long h, int[] tu, int[] tj, int[] tctx, double[] tr, int[] su, int[] sj,
                                                 int[] sctx, double[] sr, double binThold, int numRecs, int numIgnore, int strategy);
}

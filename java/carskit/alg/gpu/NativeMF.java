// NativeMF.java -- Java side of the JNI binding to libcarskit_mi355x.so (include/carskit_mi355x.h).
// Source only: the build image and the GPU box have no JDK, so this file is NOT compiled or tested here
// (see INTEGRATION.md).  One static native method per C-ABI entry point, same argument meaning.
package carskit.alg.gpu;

public final class NativeMF {
    static { System.loadLibrary("carskit_mi355x_jni"); }

    public static final int BIASEDMF = 0, CAMF_C = 1, CAMF_CI = 2, CAMF_CU = 3, CAMF_CUCI = 4;
    public static final int P = 0, Q = 1, USER_BIAS = 2, ITEM_BIAS = 3, COND_BIAS = 4, UC_BIAS = 5, IC_BIAS = 6;
    public static final int FLAG_STATE_F64 = 1, FLAG_SCHED_SERIAL = 2, FLAG_STRICT = 4, FLAG_NO_GRAPH = 16;

    /** cmi_create; returns the handle, throws RuntimeException(cmi_last_error) on failure. */
    public static native long create(int model, int k, int nUsers, int nItems, int nConds, int device, int flags);
    public static native void destroy(long h);
    /** cmi_set_ratings: rowPtr/colInd/data are the live CSR arrays of the librec SparseMatrix
     *  (getRowPointers/getColumnIndices/getData); uiUser/uiItem map a row (user-item pair id) to user/item. */
    public static native void setRatingsCsr(long h, int[] rowPtr, int[] colInd, double[] data, int[] uiUser,
                                            int[] uiItem, int[] ctxPtr, int[] ctxConds);
    /** cmi_set_state / cmi_get_state for a DenseMatrix (double[][] rows flattened by the shim) or DenseVector. */
    public static native void setMatrix(long h, int which, double[][] rows);
    public static native void getMatrix(long h, int which, double[][] rows);
    public static native void setVector(long h, int which, double[] v);
    public static native void getVector(long h, int which, double[] v);
    public static native void setHparams(long h, double regU, double regI, double regB, double regC, double globalMean);
    /** cmi_train_epoch: one pass of the for(MatrixEntry me : trainMatrix) body; returns loss (already *0.5). */
    public static native double trainEpoch(long h, double lRate);
    /** cmi_eval_ratings: {MAE, RMSE, NMAE, rMAE, rRMSE, count}. */
    public static native double[] evalRatings(long h, int[] u, int[] j, int[] ctx, double[] r, double minRate, double maxRate);
}

// CAMF_CI_GPU.java -- drop-in for carskit.alg.cars.adaptation.dependent.dev.CAMF_CI: identical constructor,
// initModel(), predict() and evalRatings() (inherited); only buildModel() is replaced.  Registered in the
// reference's factory switch next to "camf_ci" (src/carskit/main/CARSKit.java:702) as "camf_ci_gpu".
// Source only (no JDK in this image): NOT compiled or tested here.  The siblings CAMF_CU_GPU, CAMF_CUCI_GPU,
// CAMF_C_GPU (FLAG_SCHED_SERIAL) and BiasedMF_GPU differ only in the model id and the containers copied.
package carskit.alg.gpu;

import carskit.alg.cars.adaptation.dependent.dev.CAMF_CI;
import carskit.data.structure.SparseMatrix;
import java.util.List;

public class CAMF_CI_GPU extends CAMF_CI {

    public CAMF_CI_GPU(SparseMatrix trainMatrix, SparseMatrix testMatrix, int fold) {
        super(trainMatrix, testMatrix, fold);
        this.algoName = "CAMF_CI_GPU";
    }

    @Override
    protected void buildModel() throws Exception {
        // fold -> GPU round robin (the reference runs one thread per fold, CARSKit.java:395-412)
        int device = Math.max(0, fold - 1) % Math.max(1, Integer.getInteger("carskit.gpus", 1));
        long h = NativeMF.create(NativeMF.CAMF_CI, numFactors, numUsers, numItems, numConditions, device, 0);
        try {
            // getConditions(ctx) for every context id, flattened once (ContextRecommender.java:53-61)
            int numCtx = trainMatrix.numColumns();
            int[] ctxPtr = new int[numCtx + 1];
            java.util.ArrayList<Integer> conds = new java.util.ArrayList<>();
            for (int c = 0; c < numCtx; c++) {
                List<Integer> cs = getConditions(c);
                conds.addAll(cs);
                ctxPtr[c + 1] = conds.size();
            }
            int[] ctxConds = new int[conds.size()];
            for (int i = 0; i < ctxConds.length; i++) ctxConds[i] = conds.get(i);
            int numUI = trainMatrix.numRows();
            int[] uiUser = new int[numUI], uiItem = new int[numUI];
            for (int ui = 0; ui < numUI; ui++) {
                uiUser[ui] = rateDao.getUserIdFromUI(ui);
                uiItem[ui] = rateDao.getItemIdFromUI(ui);
            }
            NativeMF.setRatingsCsr(h, trainMatrix.getRowPointers(), trainMatrix.getColumnIndices(),
                                   trainMatrix.getData(), uiUser, uiItem, ctxPtr, ctxConds);
            NativeMF.setHparams(h, regU, regI, regB, regC, globalMean);
            // copy-in: Java owns the containers (SURVEY 8b "Ownership"); rows are separate heap arrays
            NativeMF.setMatrix(h, NativeMF.P, Rows.of(P));
            NativeMF.setMatrix(h, NativeMF.Q, Rows.of(Q));
            NativeMF.setVector(h, NativeMF.USER_BIAS, userBias.getData());
            NativeMF.setMatrix(h, NativeMF.IC_BIAS, Rows.of(icBias));

            for (int iter = 1; iter <= numIters; iter++) {
                loss = NativeMF.trainEpoch(h, lRate);     // replaces CAMF_CI.java:79-123
                if (isConverged(iter)) break;             // unchanged Java: IterativeRecommender.java:145-199
            }

            // copy-back so predict()/evalRatings()/saveModel() keep working unchanged
            NativeMF.getMatrix(h, NativeMF.P, Rows.of(P));
            NativeMF.getMatrix(h, NativeMF.Q, Rows.of(Q));
            NativeMF.getVector(h, NativeMF.USER_BIAS, userBias.getData());
            NativeMF.getMatrix(h, NativeMF.IC_BIAS, Rows.of(icBias));
        } finally {
            NativeMF.destroy(h);
        }
    }
}

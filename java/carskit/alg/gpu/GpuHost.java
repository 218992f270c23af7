// GpuHost.java -- what GpuSupport.buildModel needs from a *_GPU recommender.  The reference keeps its model containers and
// hyper-parameters in PROTECTED fields of carskit.generic.Recommender / IterativeRecommender / ContextRecommender, and every
// drop-in must extend a different reference class (CAMF_CI, CAMF_CU, ..., to inherit its initModel/predict), so the shared
// flow lives in GpuSupport and reaches those fields through this interface.  No JDK here: not compiled by javac; executed under the Java-source interpreter (tests/test_java_binding_exec.py).
package carskit.alg.gpu;

import carskit.data.structure.SparseMatrix;
import java.util.List;

interface GpuHost {
    int modelId();                       // NativeMF.CAMF_CI ...
    int createFlags();                   // e.g. NativeMF.FLAG_SCHED_SERIAL for CAMF_C
    int factors();
    int users();
    int items();
    int conditions();
    int foldId();
    SparseMatrix contextualTrain();      // trainMatrix (user-item pairs x contexts)
    SparseMatrix contextualTest();       // testMatrix, may be null
    librec.data.SparseMatrix train2D();  // `train` (users x items); only read by the 2-D models
    List<Integer> conditionsOf(int ctx); // ContextRecommender.getConditions
    double[] regularizers();             // {regU, regI, regB, regC} as doubles (Java float fields promoted)
    double mean();                       // globalMean
    double minRating();
    double maxRating();
    int iterations();                    // numIters
    double learnRate();                  // lRate, re-read every epoch: isConverged() -> updateLRate() changes it
    boolean evaluatesDuringTraining();   // earlyStopMeasure is MAE / RMSE / ...: isConverged() calls evalRatings() per epoch
    void prepare(long h);                // calls that must precede the ratings (CAMF_ICS/LCS/MCS: NativeMF.setSimParams); usually empty
    void copyIn(long h);                 // Dev.setMatrix / setVector of the containers this model owns
    void copyOut(long h);                // ... and back
    void handle(long h);                 // the live native handle (0 outside buildModel); evalRatings() consults it
    boolean epochDone(int iter, double epochLoss) throws Exception; // { loss = epochLoss; return isConverged(iter); }
}

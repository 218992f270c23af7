/*
 * carskit_oracle.h -- CPU restatement (fp64, single thread, order-exact) of the CARSKit
 * SGD training path.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY: the reference (irecsys/CARSKit v0.4.0, Java) ships no tests, golden vectors or
 * expected-output files for this path and there is no JVM in the build container.  Rounds 1-2
 * ran "parity unpinned"; since round 3 the restatement is pinned by INTERPRETED EXECUTION of the
 * reference itself -- not by a JVM run:
 *   (a) librec's DenseMatrix / SparseMatrix / Randoms executed from the vendored jar's bytecode
 *       (oracle/jvm/interp.py -> tests/golden/librec_l0.json, tests/test_librec_l0.py),
 *   (b) buildModel / predict / isConverged / updateLRate / evalRatings of all ten SGD recommenders
 *       and FM executed from the reference's Java source (oracle/jvm/javasrc.py ->
 *       tests/golden/reference_src.json; tests/test_reference_src_golden.py holds this library to
 *       it bit for bit).  JDK / guava classes are stand-ins written from their specifications.
 * The earlier anchors remain:
 *   (i)  an independently written second restatement (oracle/oracle_np.py) that must agree
 *        bit-for-bit on random small problems,
 *   (ii) hand-computed single-update known answers (tests/test_oracle_known_answers.py),
 *   (iii) java.util.Random known answers from the public algorithm.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (libcarskit_mi355x.so) never links, loads or calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference
 * root).  Arithmetic is IEEE double with one rounding per Java operator: build with
 * -ffp-contract=off (see oracle/Makefile) so the C compiler forms no FMA the JVM would not.
 */
#ifndef CARSKIT_ORACLE_H
#define CARSKIT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* model kinds; numbering shared with include/carskit_mi355x.h (CMI_MODEL_*) */
enum {
    ORC_BIASEDMF = 0, /* src/carskit/alg/baseline/cf/BiasedMF.java */
    ORC_CAMF_C = 1,   /* src/carskit/alg/cars/adaptation/dependent/dev/CAMF_C.java */
    ORC_CAMF_CI = 2,  /* .../dev/CAMF_CI.java */
    ORC_CAMF_CU = 3,  /* .../dev/CAMF_CU.java */
    ORC_CAMF_CUCI = 4, /* .../dev/CAMF_CUCI.java */
    ORC_PMF = 5,       /* src/carskit/alg/baseline/cf/PMF.java */
    /* carskit_oracle_sim.c */
    ORC_SVDPP = 6,     /* src/carskit/alg/baseline/cf/SVDPlusPlus.java */
    ORC_CAMF_ICS = 7,  /* src/carskit/alg/cars/adaptation/dependent/sim/CAMF_ICS.java */
    ORC_CAMF_LCS = 8,  /* .../sim/CAMF_LCS.java */
    ORC_CAMF_MCS = 9   /* .../sim/CAMF_MCS.java */
};

/* Flat view of one recommender instance's state and training tuples.
 * Tuples are in the order the reference's MatrixIterator yields them (CRS: row = user-item
 * pair id ascending, then column = context id ascending); for BiasedMF the 2-D train matrix
 * (user ascending, item ascending) and ctx is ignored.
 * ctx_ptr/ctx_conds: CSR restatement of ContextRecommender.getConditions
 * (src/carskit/generic/ContextRecommender.java:53-61): conditions of context c are
 * ctx_conds[ctx_ptr[c] .. ctx_ptr[c+1]) in the order of the reference's comma-joined key. */
typedef struct {
    int32_t model;
    int32_t k;
    int32_t n_users, n_items, n_conds;
    int64_t n;                /* training tuples */
    const int32_t *u, *j, *ctx;
    const double *r;
    const int32_t *ctx_ptr;   /* n_ctx+1, may be NULL for BiasedMF */
    const int32_t *ctx_conds;
    double *P;                /* n_users x k row-major */
    double *Q;                /* n_items x k row-major */
    double *userBias;         /* n_users            (BiasedMF, CAMF_C, CAMF_CI) */
    double *itemBias;         /* n_items            (BiasedMF, CAMF_C, CAMF_CU) */
    double *condBias;         /* n_conds            (CAMF_C) */
    double *ucBias;           /* n_users x n_conds  (CAMF_CU, CAMF_CUCI) */
    double *icBias;           /* n_items x n_conds  (CAMF_CI, CAMF_CUCI) */
    double globalMean;
    double regU, regI, regB, regC; /* Java floats promoted to double (IterativeRecommender.java:40) */
} orc_problem;

/* Learning-rate schedule + convergence state (IterativeRecommender.java:56-72). */
typedef struct {
    double lRate;        /* starts at (double)initLRate, IterativeRecommender.java:106 */
    double maxLRate;     /* (double)(float) -max, <=0 = unlimited */
    double decay;        /* (double)(float) -decay, outside (0,1) = off */
    int32_t boldDriver;
    int32_t earlyStop;   /* 0 none, 1 Loss ; MAE/RMSE handled by the caller via measure */
    double loss, last_loss;
    double measure, last_measure;
} orc_schedule;

/* one pass of buildModel()'s for(MatrixEntry me : trainMatrix) body; returns loss*0.5 */
double orc_sgd_epoch(const orc_problem *p, double lRate);

/* IterativeRecommender.isConverged + updateLRate (IterativeRecommender.java:145-229).
 * s->loss must hold this epoch's loss; if earlyStop is MAE/RMSE the caller stores the
 * measure in s->measure first and passes use_measure=1.  Returns 1 if converged, -1 if the
 * loss is NaN/Inf (the reference calls System.exit(-1) there), else 0. */
int orc_is_converged(orc_schedule *s, int iter, int use_measure);

/* whole buildModel(): up to numIters epochs; returns the number of epochs run.  losses[] and
 * lrates[] (each numIters long, may be NULL) receive the per-epoch loss and the lRate used. */
int orc_build_model(const orc_problem *p, orc_schedule *s, int numIters, double *losses, double *lrates);

/* predict(u,j,c) of the model (unbounded) */
double orc_predict(const orc_problem *p, int32_t u, int32_t j, int32_t ctx);
/* ranking(u, j, c) for a list of items (Recommender.java:806: one predict() per candidate) */
void orc_predict_items(const orc_problem *p, int32_t u, int32_t ctx, int32_t n, const int32_t *items, double *out);

/* Recommender.evalRatings numeric part (src/carskit/generic/Recommender.java:504-594).
 * out[0]=MAE out[1]=RMSE out[2]=NMAE out[3]=rMAE out[4]=rRMSE ; returns numCount */
int64_t orc_eval_ratings(const orc_problem *p, int64_t n_test, const int32_t *tu, const int32_t *tj,
                         const int32_t *tctx, const double *tr, double minRate, double maxRate,
                         double *out, double *preds /* n_test or NULL */);

/* SparseMatrix.getGlobalAvg (src/carskit/data/structure/SparseMatrix.java:49-56):
 * sequential sum of stored values / number of non-zero stored values */
double orc_global_mean(const double *r, int64_t n);

/* java.util.Random (public algorithm) */
typedef struct {
    uint64_t seed;
    double nextNextGaussian;
    int32_t haveNextNextGaussian;
} orc_jrandom;
void orc_jrandom_seed(orc_jrandom *g, int64_t seed);
int32_t orc_jrandom_next(orc_jrandom *g, int bits);
double orc_jrandom_next_double(orc_jrandom *g);
double orc_jrandom_next_gaussian(orc_jrandom *g);
/* librec DenseMatrix.init(mean,sigma) / init() restated over a flat array */
void orc_init_gaussian(orc_jrandom *g, double *a, int64_t n, double mean, double sigma);
void orc_init_uniform(orc_jrandom *g, double *a, int64_t n, double range);

/* ---- SVD++ and the similarity-based CAMF family, see carskit_oracle_sim.c ------------------------
 * SVD++ iterates the 2-D train matrix (u, j, r; ctx unused) and needs userItemsCache = the items of every user in that matrix
 * as CSR (ui_ptr / ui_items, items ascending = librec rowColumnsCache).  The CAMF_*CS models iterate the contextual matrix and
 * pair the i-th condition of a context with empty_conds[i] (EmptyContextConditions, the ":na" conditions in header order). */
typedef struct {
    int32_t model, k, n_users, n_items, n_conds, numF;
    int64_t n;
    const int32_t *u, *j, *ctx;
    const double *r;
    const int32_t *ctx_ptr, *ctx_conds, *empty_conds;
    const int32_t *ui_ptr, *ui_items;  /* SVD++ */
    double *P, *Q;                     /* n_users x k, n_items x k */
    double *userBias, *itemBias, *Y;   /* SVD++: Y is n_items x k */
    double *ccMatrix;                  /* CAMF_ICS: n_conds x n_conds, kept symmetric (librec SymmMatrix) */
    double *cfMatrix;                  /* CAMF_LCS: n_conds x numF */
    double *cVector;                   /* CAMF_MCS: n_conds */
    double globalMean, regU, regI, regB, regC;
    double upbound, lowbound;          /* CAMF_MCS.java:47-48: 1/sqrt(numContextDims), 1e-100 */
} orc_sim_problem;
double orc_sim_predict(const orc_sim_problem *p, int32_t u, int32_t j, int32_t ctx);
double orc_sim_epoch(const orc_sim_problem *p, double lRate); /* one pass of buildModel()'s loop, loss already scaled */

/* ---- FM (src/carskit/alg/cars/adaptation/dependent/FM.java), see carskit_oracle_fm.c ---------------- */
typedef struct {
    int32_t k, n_users, n_items, n_conds, n_ctx_dims;
    int32_t p;             /* n_users + n_items + n_conds (FM.java:62) */
    int64_t size;          /* training tuples, CRS order */
    const int32_t *u, *j, *ctx;
    const double *r;
    double w0;
    double *w;             /* p */
    double *V;             /* p x k row-major */
    double *Q;             /* size x k row-major (FM.java:72) */
    double *errors;        /* size */
    double regLw, regLf;   /* Java floats promoted (FM.java:53-54) */
} orc_fm;
double orc_fm_predict(const orc_fm *m, int32_t u, int32_t j, int32_t c);
void orc_fm_init(orc_fm *m);
double orc_fm_sweep(orc_fm *m);

#ifdef __cplusplus
}
#endif
#endif

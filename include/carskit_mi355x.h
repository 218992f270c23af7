/*
 * carskit_mi355x.h -- C ABI of libcarskit_mi355x.so: the MI355X (gfx950) replacement for the
 * per-rating SGD inner loop of buildModel() in CARSKit's BiasedMF / CAMF_C / CAMF_CI / CAMF_CU /
 * CAMF_CUCI recommenders, plus the numeric part of evalRatings().
 *
 * The reference (irecsys/CARSKit v0.4.0) is Java and has no FFI of its own; this is the surface a
 * JNI shim (see INTEGRATION.md and java/) binds.  Every entry point names the reference code it
 * stands in for (paths relative to the reference root).  Plain pointers and sizes only; the
 * caller owns every host buffer; the library owns the device copies.
 *
 * Threading: a handle is an independent recommender instance bound to one GPU; different handles
 * may be used from different threads concurrently (the reference runs one Java thread per CV
 * fold, src/carskit/main/CARSKit.java:395-412).  One handle must not be used from two threads at
 * once.  No process-global state.
 *
 * Errors: every function returns CMI_OK (0) or a negative CMI_E_* code; cmi_last_error() gives
 * the message (the reference throws checked Exceptions, src/carskit/generic/Recommender.java:319;
 * the JNI shim turns a non-zero status into a RuntimeException).  There is NO CPU fallback: without
 * a HIP device every compute entry point fails with CMI_E_NO_DEVICE.
 */
#ifndef CARSKIT_MI355X_H
#define CARSKIT_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMI_ABI_VERSION 5 /* 5: round 6 -- CMI_E_BUSY; cmi_group_exchange_path (RCCL pre-flight + fallback); FM: the fixed-order (bit-reproducible) sweep is the DEFAULT, CMI_FM_FLAG_RELAXED_SUMS opts into the
                             LDS-atomic form (CMI_FM_FLAG_DETERMINISTIC is still accepted and now changes nothing)
                             4: round 5 -- ADDED cmi_group_last_times, cmi_comm_last_exchange_ms (exchange vs compute time of an epoch),
                             cmi_chain_schedule_device; CMI_E_HOST; CMI_FM_FLAG_DETERMINISTIC; cmi_fm_layout's [5..6] are batches
                             3: round 4 -- cmi_comm_*, cmi_fm_comm_*, group resident evaluation, FM layout / timing, ranking host
                             clock; REMOVED: CMI_FLAG_SCHED_FLOW, CMI_FLAG_TWO_LANE, cmi_flow_schedule, cmi_split_schedule */

/* status codes */
#define CMI_OK 0
#define CMI_E_INVALID (-1)    /* bad argument / bad call order */
#define CMI_E_NO_DEVICE (-2)  /* no HIP device (no CPU fallback exists) */
#define CMI_E_HIP (-3)        /* a HIP runtime call failed */
#define CMI_E_NUMERIC (-4)    /* loss became NaN/Inf (IterativeRecommender.java:181-184) */
#define CMI_E_UNSUPPORTED (-5)
#define CMI_E_HOST (-6)       /* a C++ exception (std::bad_alloc ...) reached the boundary: caught there, never thrown at the host;
                                 the JNI shim turns it into a RuntimeException like every other status (Recommender.java:1162-1171) */
#define CMI_E_BUSY (-7)       /* an owner (persistent) epoch was NOT launched: another process held the device's owner-epoch lock for the
                                 whole bounded wait, or the lock file cannot be opened (CMI_OWNER_NO_LOCK=1 waives it); model untouched */

/* recommender kinds = the `recommender=` names the reference's factory switch maps to the classes
 * this library accelerates (src/carskit/main/CARSKit.java:461,700-707) */
#define CMI_MODEL_BIASEDMF 0  /* src/carskit/alg/baseline/cf/BiasedMF.java */
#define CMI_MODEL_CAMF_C 1    /* src/carskit/alg/cars/adaptation/dependent/dev/CAMF_C.java */
#define CMI_MODEL_CAMF_CI 2   /* .../dev/CAMF_CI.java */
#define CMI_MODEL_CAMF_CU 3   /* .../dev/CAMF_CU.java */
#define CMI_MODEL_CAMF_CUCI 4 /* .../dev/CAMF_CUCI.java */
#define CMI_MODEL_PMF 5       /* src/carskit/alg/baseline/cf/PMF.java:47-82: BiasedMF without biases and without
                                 globalMean (predict = rowMult, IterativeRecommender.java:126-128); 2-D train matrix */
/* the remaining SGD recommenders of the same family (CARSKit.java:469,708-712).  Every rating updates parameters every other
 * rating reads (Y rows of all the user's items; the condition-similarity scalars / vectors / positions), so their exact semantics
 * are one dependent chain: CMI_FLAG_SCHED_SERIAL is required, as for CAMF_C.  cmi_set_sim_params must be called before
 * cmi_set_ratings for the three CAMF_*CS models. */
#define CMI_MODEL_SVDPP 6     /* src/carskit/alg/baseline/cf/SVDPlusPlus.java:46-146 (2-D train matrix, like BiasedMF) */
#define CMI_MODEL_CAMF_ICS 7  /* src/carskit/alg/cars/adaptation/dependent/sim/CAMF_ICS.java:33-133 */
#define CMI_MODEL_CAMF_LCS 8  /* .../sim/CAMF_LCS.java:34-147 */
#define CMI_MODEL_CAMF_MCS 9  /* .../sim/CAMF_MCS.java:38-166 */

/* state containers = the model fields a subclass must leave consistent
 * (IterativeRecommender.java:56-64, CAMF.java:40-42) */
#define CMI_STATE_P 0         /* DenseMatrix P        numUsers x k */
#define CMI_STATE_Q 1         /* DenseMatrix Q        numItems x k */
#define CMI_STATE_USER_BIAS 2 /* DenseVector userBias numUsers */
#define CMI_STATE_ITEM_BIAS 3 /* DenseVector itemBias numItems */
#define CMI_STATE_COND_BIAS 4 /* DenseVector condBias numConditions         (CAMF_C) */
#define CMI_STATE_UC_BIAS 5   /* DenseMatrix ucBias   numUsers x numConditions (CAMF_CU, CAMF_CUCI) */
#define CMI_STATE_IC_BIAS 6   /* DenseMatrix icBias   numItems x numConditions (CAMF_CI, CAMF_CUCI) */
#define CMI_STATE_Y 7         /* DenseMatrix Y            numItems x k                (SVD++, SVDPlusPlus.java:36) */
#define CMI_STATE_CC_MATRIX 8 /* SymmMatrix ccMatrix_ICS  numConditions x numConditions, exchanged as the full symmetric matrix (CAMF.java:45) */
#define CMI_STATE_CF_MATRIX 9 /* DenseMatrix cfMatrix_LCS numConditions x numF       (CAMF.java:46; numF from cmi_set_sim_params) */
#define CMI_STATE_C_VECTOR 10 /* DenseVector cVector_MCS  numConditions               (CAMF.java:47) */
#define CMI_STATE_COUNT 11

/* host buffer element types for cmi_set_state / cmi_get_state */
#define CMI_DTYPE_F32 0
#define CMI_DTYPE_F64 1

/* cmi_create flags.  CAMF_C updates the condBias vector shared by every tuple, so its tuples do not
 * commute and it requires CMI_FLAG_SCHED_SERIAL (cmi_create returns CMI_E_UNSUPPORTED otherwise). */
#define CMI_FLAG_STATE_F64 0x1u  /* keep the model in fp64 on the GPU (reference precision); default fp32 */
#define CMI_FLAG_SCHED_SERIAL 0x2u /* one wavefront walks the tuples in the reference's CRS order (exact for
                                      every model incl. CAMF_C; slow).  Default: dependency-level schedule --
                                      tuples that share no user and no item commute exactly, so the result equals
                                      the sequential order at equal precision while whole levels run in parallel */
#define CMI_FLAG_STRICT 0x4u     /* generic kernel + strictly left-to-right dot product (DenseMatrix.rowMult
                                      order) and, under SCHED_SERIAL, the reference's running-sum order for `loss`:
                                      with STATE_F64 the model (and under SERIAL the loss) is bit-identical to the
                                      Java arithmetic */
/* 0x20u, 0x40u: CMI_FLAG_SCHED_FLOW / CMI_FLAG_TWO_LANE of ABI 2 (round-1 experiments, measured slower than the level launches)
 * were removed in ABI 3; the bits are reserved */
#define CMI_FLAG_SCHED_CHAIN 0x80u /* force the hub-chain level schedule (the default whenever its levels are wide enough):
                                      consecutive tuples of one item (or user) whose other row is already final run back to
                                      back in one 16-lane group with the shared row, its bias and its context-bias row kept on
                                      chip; same result as the plain level schedule, bit for bit in fp32.
                                      CMI_E_UNSUPPORTED at cmi_set_ratings if the model/k has no chain kernel */
#define CMI_FLAG_NO_CHAIN 0x100u /* never use the hub-chain schedule (A/B runs against the plain level launches) */
#define CMI_FLAG_SCHED_OWNER 0x200u /* heavy-tailed degrees: ONE persistent launch per epoch in which every row of the heavy side
                                     * (items, or users) is owned by one wavefront that walks the row's tuples in CRS order with the
                                     * row in registers; the other side's rows travel between owners as tagged records
                                     * (owner_kernels.hip).  Order-exact like the level schedules; k <= 256 (fp64: 128), <= 384 conditions */
#define CMI_FLAG_NO_OWNER 0x400u /* never pick the owner schedule automatically (it is picked for >= 2^16 tuples whose dependency levels
                                  * are narrow -- heavy-tailed degrees -- when its estimated epoch is at least twice shorter) */
#define CMI_FLAG_SPOKE_ARENA 0x800u /* hub-chain schedule: keep the spoke rows in an arena of one slot per tuple, in schedule order -- the row
                                      of the tuple at stream position p is READ from slot p (a level's spoke reads become one sequential
                                      stream) and WRITTEN to the slot of the same row's next tuple.  Same arithmetic, same bits; costs
                                      tuples x k x 4 (8) bytes of HBM.  Picked automatically when the spoke table is >= 2 GiB (random 512-B
                                      rows over such a table run at 0.5 of the HBM peak, the arena form at 0.7) and the arena fits in 60 % of
                                      the free device memory; CMI_FLAG_NO_ARENA opts out */
#define CMI_FLAG_NO_ARENA 0x1000u
#define CMI_FLAG_NO_GRAPH 0x10u  /* launch the per-level kernels eagerly instead of replaying a hipGraph */

typedef struct cmi_instance *cmi_handle;

/* library / device probes (no reference counterpart) */
int cmi_abi_version(void);
int cmi_device_count(void); /* 0 when no HIP device is visible; never fails */
/* measurement: what this GPU sustains right now, GB/s (read + write bytes) over a scratch buffer of `bytes` (>= 1 MiB; use >= 2 GiB
 * to be past the 256 MB Infinity Cache): out[0] = streaming float4 copy, out[1] = random 512-byte-row read-modify-write (the SGD
 * kernels' row traffic pattern).  bench.py reports both beside the spec peak. */
int cmi_measure_hbm(int device, int64_t bytes, double out[2]);

/* new Recommender(trainMatrix, testMatrix, fold) + initModel() container allocation
 * (Recommender.java:180-275, IterativeRecommender.java:232-247, CAMF_CI.java:51-63 ...).
 * State is zero until cmi_set_state: the library never draws random numbers (the reference's
 * init stream is unseedable), the host injects the arrays. */
int cmi_create(int model, int k, int n_users, int n_items, int n_conds, int device, unsigned flags,
               cmi_handle *out);
int cmi_destroy(cmi_handle h);
/* message of the last failure on this handle; h == NULL: last cmi_create failure on this thread */
const char *cmi_last_error(cmi_handle h);

/* The training matrix: what `for (MatrixEntry me : trainMatrix)` yields, flattened
 * (CAMF_CI.java:80-86; librec SparseMatrix CSR arrays + DataDAO.getUserIdFromUI/getItemIdFromUI,
 * src/carskit/data/processor/DataDAO.java:1038-1046).  n tuples in CRS order: u[t], j[t] inner
 * user/item ids, ctx[t] context-combination id, r[t] rating.  ctx_ptr/ctx_conds (n_ctx+1 / nnz) is
 * ContextRecommender.getConditions (src/carskit/generic/ContextRecommender.java:53-61) as CSR.
 * BiasedMF: the 2-D `train` matrix (BiasedMF.java:62-66); ctx, ctx_ptr, ctx_conds may be NULL.
 * Builds the execution schedule (host integer work) and uploads. */
int cmi_set_ratings(cmi_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx,
                    const double *r, int32_t n_ctx, const int32_t *ctx_ptr, const int32_t *ctx_conds);

/* copy-in at buildModel() entry / copy-back at exit of one model container (row-major, count
 * elements = rows*cols).  Converts between the host dtype and the device state dtype. */
int cmi_set_state(cmi_handle h, int which, const void *src, int64_t count, int dtype);
int cmi_get_state(cmi_handle h, int which, void *dst, int64_t count, int dtype);

/* CAMF_ICS / CAMF_LCS / CAMF_MCS: EmptyContextConditions (ContextRecommender.java:43 = the ":na" conditions in header order,
 * DataDAO.java:213-214; the i-th condition of a context is compared with empty_conds[i]), numF = `-f` of the CAMF_LCS option line
 * (CAMF_LCS.java:37; allocates cfMatrix_LCS), and rateDao.numContextDims() (CAMF_MCS.java:47: upbound = 1/sqrt(dims)).  Ignored
 * by the other models. */
int cmi_set_sim_params(cmi_handle h, int num_f, int n_ctx_dims, const int32_t *empty_conds, int n_empty);

/* regU/regI/regB/regC (IterativeRecommender.java:40,94-98: Java floats promoted to double) and
 * globalMean (Recommender.java:265) */
int cmi_set_hparams(cmi_handle h, double regU, double regI, double regB, double regC, double global_mean);

/* `cv -p on` (CARSKit.java:395-412: one thread per fold, every fold a recommender of its own): `instances` = how many instances the
 * host trains CONCURRENTLY on this handle's device.  A hint, not semantics: the persistent kernels (the owner epoch, one launch whose
 * wavefronts must all be resident) size their grids to 1 / instances of the device, so the instances' launches run side by side instead
 * of one after another; results are what they are without the hint (the schedules are order-exact).  Call before cmi_set_ratings.
 * Default 1.  instances < 1 -> CMI_E_INVALID. */
int cmi_set_device_share(cmi_handle h, int instances);

/* One pass of the `for (MatrixEntry me : trainMatrix) {...}` body + `loss *= 0.5`
 * (e.g. CAMF_CI.java:79-123) with learning rate lrate; *loss_out = the epoch's loss.  The host keeps
 * calling isConverged() itself (the Java drop-in does exactly this). */
int cmi_train_epoch(cmi_handle h, double lrate, double *loss_out);

/* Whole buildModel(): up to num_iters epochs with IterativeRecommender.isConverged/updateLRate
 * (IterativeRecommender.java:145-229) evaluated between epochs.  early_stop: 0 none, 1 loss.
 * losses/lrates (num_iters each, may be NULL) receive the per-epoch loss and the rate used;
 * *iters_run the number of epochs executed; *final_lrate (may be NULL) the lRate field afterwards.
 * NaN/Inf loss -> CMI_E_NUMERIC (the reference calls System.exit(-1)). */
int cmi_train(cmi_handle h, int num_iters, double init_lrate, double max_lrate, int bold_driver, double decay,
              int early_stop, double *losses, double *lrates, int *iters_run, double *final_lrate);

/* saveModel() / loadModel() (IterativeRecommender.java:249-292), as ONE documented, versioned, checksummed file holding every
 * container of the model (the reference serialises P, Q, userBias, itemBias with Java object streams and forgets condBias /
 * ucBias / icBias), the hyper-parameters of cmi_set_hparams, and the resume state of the epoch loop (the learning rate the next
 * epoch would use, the last epoch's loss, epochs done): training N epochs, saving, loading into a fresh handle and training M more
 * equals training N + M epochs.  Layout: carskit_amd/csrc/model_io.cpp.  cmi_load_model checks that the file matches the handle
 * (model, k, sizes) and its checksum before it changes anything; lrate / last_loss / epochs_done may be NULL.  Host I/O only (the
 * state moves through cmi_get_state / cmi_set_state).  NOT the reference's Java serialization format (no JVM to verify one). */
int cmi_save_model(cmi_handle h, const char *path, double lrate, double last_loss, int epochs_done);
int cmi_load_model(cmi_handle h, const char *path, double *lrate, double *last_loss, int *epochs_done);

/* cmi_train continued: epochs first_iter .. first_iter + num_iters - 1 of the same loop, prev_loss = the loss of epoch
 * first_iter - 1 (what isConverged()/updateLRate() compare with).  cmi_train == cmi_train_from(first_iter = 1, prev_loss = 0).
 * With cmi_save_model / cmi_load_model: N epochs, save, load, cmi_train_from(N + 1, ...) equals N + M epochs in one go. */
int cmi_train_from(cmi_handle h, int first_iter, double prev_loss, int num_iters, double init_lrate, double max_lrate,
                   int bold_driver, double decay, int early_stop, double *losses, double *lrates, int *iters_run,
                   double *final_lrate);

/* predict(u, j, c, bound) for n tuples (Recommender.java:306-317 + the model's predict());
 * bound != 0 clamps to [lo, hi].  ctx may be NULL for BiasedMF. */
int cmi_predict_batch(cmi_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, int bound,
                      double lo, double hi, double *out);

/* numeric part of Recommender.evalRatings (Recommender.java:504-594) over n test tuples whose ctx
 * ids index the SAME ctx_ptr/ctx_conds table given to cmi_set_ratings.
 * out[0]=MAE out[1]=RMSE out[2]=NMAE out[3]=rMAE out[4]=rRMSE; *count = tuples with non-NaN prediction */
int cmi_eval_ratings(cmi_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx,
                     const double *r, double min_rate, double max_rate, double *out, int64_t *count);

/* ---- plumbing for the host layer (multi-GPU exchange, measurement); no reference counterpart ---- */

/* `--early-stop MAE|RMSE` evaluates the test set after every epoch (IterativeRecommender.java:156-161): upload the test tuples
 * once, then evaluate from device memory.  Same numbers as cmi_eval_ratings on the same tuples. */
int cmi_set_eval_ratings(cmi_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, const double *r);
int cmi_eval_resident(cmi_handle h, double min_rate, double max_rate, double out[5], int64_t *count);

/* ---- top-N ranking evaluation: Recommender.evalRankings (src/carskit/generic/Recommender.java:668-964) ----------
 * The reference scores every candidate item for every test (user, context) pair with one predict() call each
 * (O(queries x items x k)); here that scoring is one dense contraction per query batch plus a fused top-N selection
 * on the GPU, and only the per-query metric formulas (happy.coding.math.Measures via src/carskit/eval/Measures.java)
 * run on the host.
 *   train tuples  -> candidate items = items seen in training, in java.util.HashSet<Integer> iteration order
 *                    (DataDAO.getItemList, DataDAO.java:1210-1218); minus the num_ignore most-rated ones
 *                    (Recommender.java:720-735); and the per-(user, context) already-rated items that are skipped
 *                    (Recommender.java:793-816).  tr may be NULL (all entries non-zero).
 *   test tuples   -> queries: (user, context) pairs with at least one item rated > bin_thold that is a candidate
 *                    (DataDAO.getUserCtxList(sm, threshold), DataDAO.java:1114-1140; Recommender.java:776-790).
 *   score         -> ranking(u,j,c) = predict(u,j,c) unbounded (Recommender.java:1016-1018); kept if not NaN and
 *                    > bin_thold; sorted descending, ties in candidate order (stable Collections.sort); cut at num_recs.
 *   num_recs      -> `-topN`; must be >= 1 (the reference throws for a negative cut-off, Measures.java:13-16).
 *   strategy      -> CMI_RANK_UCU: mean over contexts per user, then over users; CMI_RANK_UC: mean over all pairs
 *                    (`evaluation.setup --rand-seed ... -strategy`, Recommender.java:856-926).  Stats.mean skips NaN.
 * out[21] = Pre5 Pre10 PreN Rec5 Rec10 RecN AUC5 AUC10 AUCN MAP5 MAP10 MAPN NDCG5 NDCG10 NDCGN MRR5 MRR10 MRRN D5 D10 DN
 * (D* = 0: diversity is off unless `-diverse`, which needs the item-similarity cache and is not built).
 * Optional per-query outputs (NULL to skip), sized for n_test queries: q_user/q_ctx/q_count[n] and
 * top_items/top_scores[n * num_recs] (inner item ids, -1 / NaN padded) -- the `-isResultsOut` list of the reference.
 * Context ids index the ctx table given to cmi_set_ratings (2-D models ignore the table). */
#define CMI_RANK_MEASURES 21
#define CMI_RANK_UCU 0
#define CMI_RANK_UC 1
int cmi_eval_rankings(cmi_handle h, int64_t n_train, const int32_t *tu, const int32_t *tj, const int32_t *tctx,
                      const double *tr, int64_t n_test, const int32_t *su, const int32_t *sj, const int32_t *sctx,
                      const double *sr, double bin_thold, int num_recs, int num_ignore, int strategy,
                      double out[CMI_RANK_MEASURES], int64_t *n_queries, int32_t *q_user, int32_t *q_ctx,
                      int32_t *q_count, int32_t *top_items, double *top_scores);
/* measurement: GPU time (HIP events on cmi_stream()) of the most recent cmi_eval_rankings scoring loop (operand
 * gather + contraction + mask + top-N over all query batches) and the flops of its contraction
 * (2 x queries x candidates x padded operand length) */
int cmi_last_rank_ms(cmi_handle h, float *ms, double *flops);
/* host wall clock of the last cmi_eval_rankings call, milliseconds: out[0] plan (candidates, queries, exclusions), [1] setup (buffers,
 * uploads, item operands), [2] the scoring loop including the per-batch measures computed behind it, [3] tail (last batch's measures +
 * the averages), [4] the whole call */
int cmi_last_rank_host_ms(cmi_handle h, double out[5]);
/* GPU time of the last evaluation's two kernels, by HIP events around their launches inside the scoring loop, milliseconds summed over
 * the batches: out[0] the contraction (rank_gemm_mfma_f32), out[1] the selection (rank_topn_split); both 0 when the evaluation did not
 * run the split form (fp64 state, SVD++ / CAMF_ICS / LCS / MCS) */
int cmi_last_rank_kernel_ms(cmi_handle h, double out[2]);
/* host-only (no GPU): the bookkeeping cmi_eval_rankings does before scoring -- candidate items in HashSet<Integer> order minus the
 * `num_ignore` most rated (Recommender.java:704-735), the (user, context) queries with their correct items (:776-790), and per
 * query the candidate POSITIONS of the items already rated in that context (:793-816).  Call once with null arrays for
 * sizes = {n_cand, n_queries, n_truth, n_excl}, then with arrays (truth_ptr / excl_ptr hold n_queries + 1 offsets). */
int cmi_rank_plan(int32_t n_users, int32_t n_items, int64_t n_train, const int32_t *tu, const int32_t *tj, const int32_t *tctx,
                  const double *tr, int64_t n_test, const int32_t *su, const int32_t *sj, const int32_t *sctx, const double *sr,
                  double bin_thold, int num_ignore, int64_t sizes[4], int32_t *cand, int32_t *q_user, int32_t *q_ctx,
                  int64_t *truth_ptr, int32_t *truth_items, int64_t *excl_ptr, int32_t *excl_idx);
/* host-only: the 18 measures of ONE ranked list exactly as cmi_eval_rankings computes them per query (out[measure*3 + cut-off],
 * measures Pre Rec AUC MAP NDCG MRR, cut-offs {5, 10, num_recs}); ranked = the list already cut at num_recs, truth = the query's
 * correct items in ascending order, num_dropped = candidates not listed (Recommender.java:850-858) */
int cmi_rank_list_measures(const int32_t *ranked, int len, const int32_t *truth_sorted, int n_truth, int num_dropped,
                           int num_recs, double out[18]);
/* iteration order of a java.util.HashSet<Integer> after add()ing values[0..n) (the candidate-item order above) */
int cmi_java_int_hashset_order(int64_t n, const int32_t *values, int32_t *out, int64_t *n_out);

/* device pointer of a state container (element type per CMI_FLAG_STATE_F64), so the host can run its
 * epoch-boundary exchange (RCCL all-reduce of item-side deltas) in place.  The table behind the pointer is current
 * when the call returns and host writes through it are honoured by the next epoch (a container whose live rows sit
 * in the spoke arena is re-read from the table); call it again after every training call before touching the memory. */
int cmi_state_device_ptr(cmi_handle h, int which, void **ptr, int64_t *count, int *dtype);
/* the HIP stream (hipStream_t) all of this handle's work is enqueued on */
int cmi_stream(cmi_handle h, void **stream);
/* Epoch-boundary exchange of the REPLICATED item-side containers when ratings are sharded by user over several GPUs
 * (carskit_amd/dist.py; the reference loop is sequential, CAMF_CI.java:79-123, so this has no counterpart there).  One flat
 * device buffer, the "bucket" (element type = state dtype), holds container after container (Q, then itemBias / icBias as the
 * model owns them; every segment padded to 4 elements, the total to a multiple of pad_to so that it splits evenly over the
 * ranks of a reduce-scatter) this rank's movement of the item-side state since the last snapshot:
 *   cmi_exchange_setup  allocates bucket + snapshot, snapshot = current state; returns the bucket's device address / length
 *   cmi_exchange_pack   bucket = state - snapshot                       -> the host sums the bucket over ranks (RCCL)
 *   cmi_exchange_apply  state = snapshot + scale * bucket; snapshot = state        (scale = 1/W: mean of the ranks' moves)
 * All enqueued on cmi_stream().  CAMF_C (shared condBias, serial-only) is not sharded: CMI_E_UNSUPPORTED. */
int cmi_exchange_setup(cmi_handle h, int64_t pad_to, void **bucket, int64_t *count);
int cmi_exchange_pack(cmi_handle h);
int cmi_exchange_apply(cmi_handle h, double scale);
/* device address of the double the most recent epoch's loss is left in (so the host can sum it over ranks on the device) */
int cmi_loss_device_ptr(cmi_handle h, void **ptr);
int cmi_synchronize(cmi_handle h);
/* enqueue one epoch without reading the loss back (pair with cmi_synchronize / cmi_last_loss) */
int cmi_train_epoch_async(cmi_handle h, double lrate);
int cmi_last_loss(cmi_handle h, double *loss_out);
/* schedule facts: info[0]=level launches per epoch (a long run of narrow final levels counts as ONE launch: a single
 * workgroup walks them),  info[1]=largest level, info[2]=tuples,
 * info[3]=max conditions per tuple (D), info[4]=state bytes on device, info[5]=tuple-stream bytes on device,
 * info[6]=schedule kind actually running (0 level launches, 1 serial, 2 dataflow, 3 two-lane level graph,
 * 4 hub-chain levels along items, 5 hub-chain levels along users; then info[1]=most units in a level, info[7]=units;
 * 6 owner epoch with items owned, 7 with users owned; then info[0]=1, info[1]=tuples of the busiest owner, info[7]=owners in the
 * low 32 bits and, in the high 32 bits, how many of them run as teams of three wavefronts),
 * info[7]=workgroups of the dataflow launch; for CAMF_C the number of conflict-free CRS blocks its epoch is cut into
 * (0: the serial wave) */
int cmi_schedule_info(cmi_handle h, int64_t info[8]);
/* host-only: the bookkeeping of the spoke arena (CMI_FLAG_SPOKE_ARENA).  spoke[p] = spoke row id of the tuple at stream position p;
 * next[p] = position of the next tuple of the same row, the last one wrapping to the first; first[row] = position of the row's first
 * tuple or -1 */
int cmi_arena_positions(int64_t n, const int32_t *spoke, int32_t n_spokes, int32_t *next, int32_t *first);
/* "" or one sentence saying why cmi_set_ratings could not pick the schedule the data calls for (heavy-tailed degrees outside the owner
 * epoch's limits: the order-exact level walk runs, correct but roughly 10x slower) -- visible to the host instead of silent */
const char *cmi_schedule_note(cmi_handle h);
/* HBM bytes one epoch of the loaded schedule has to move, derived from the schedule (no reference counterpart: measurement).
 * out[0]: every scattered scalar billed at its 64-byte sector, read and written (hub-chain schedules: hub row, hub bias and hub
 * context-bias row once per UNIT; spoke row, tuple stream, spoke bias and spoke context-bias cells per tuple); out[1]: the same with
 * scalars at their own size; out[2]: SURVEY 8(d)'s no-reuse algorithmic bytes; out[3]: bit 0 = on-chip hub reuse is modelled, bit 1 = the spoke arena is in use */
int cmi_schedule_traffic(cmi_handle h, int64_t out[4]);
/* GPU time of the most recent epoch's kernels measured with HIP events on cmi_stream() */
int cmi_last_epoch_ms(cmi_handle h, float *ms);

/* ---- one recommender over several GPUs from ONE host process (carskit_amd/csrc/group_api.cpp) -------------------------------
 * The reference trains one fold on one thread (CARSKit.java:395-412 is its only parallelism: a thread per fold) -- there is no
 * counterpart to translate.  SURVEY 8b's `device_mask` entry, as a handle of its own: ratings are sharded BY USER over n_shards
 * instances (contiguous user ranges holding about n / n_shards tuples each; user-side containers P, userBias, ucBias live on
 * exactly one shard), the item-side containers (Q, itemBias, icBias) are replicated and merged after every epoch:
 *     item_side = snapshot + (sum over shards of (item_side_shard - snapshot)) / n_shards        (the MEAN of the shards' moves)
 * through ncclReduceScatter + ncclAllGather over xGMI on the shards' own streams (librccl, one communicator per shard, grouped
 * calls), and the fp64 epoch losses are all-reduced, so cmi_group_train_epoch returns the GLOBAL loss and the unchanged host-side
 * isConverged()/updateLRate() keeps steering.  Shards that share a device (or CMI_GROUP_NO_RCCL=1) use an in-process exchange
 * (sum in shard order on shard 0's stream) instead -- same arithmetic.  n_shards = 1 is exactly the single-instance path.
 * With n_shards > 1 this is NOT the reference's sequential semantics (n_shards local SGD streams merged per epoch): accuracy is
 * reported as a band against the 1-GPU result.  CAMF_C / SVD++ / CAMF_*CS are serial chains: 1 shard only.
 * Call order: create, set_hparams, set_ratings (creates the instances: their user counts depend on the cut), set_state, train.
 * devices: n_shards device indices (may repeat), or NULL = round robin over the visible devices. */
typedef struct cmi_group *cmi_group_handle;
int cmi_group_create(int model, int k, int n_users, int n_items, int n_conds, int n_shards, const int *devices, unsigned flags,
                     cmi_group_handle *out);
int cmi_group_destroy(cmi_group_handle g);
const char *cmi_group_last_error(cmi_group_handle g);
int cmi_group_size(cmi_group_handle g);
int cmi_group_set_hparams(cmi_group_handle g, double regU, double regI, double regB, double regC, double global_mean);
/* the WHOLE training matrix, as for cmi_set_ratings; the library cuts it by user */
int cmi_group_set_ratings(cmi_group_handle g, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, const double *r,
                          int32_t n_ctx, const int32_t *ctx_ptr, const int32_t *ctx_conds);
/* whole containers in the reference's shapes (count = rows * cols over ALL users / items): user-side ones are scattered to /
 * gathered from the owning shards, item-side ones are replicated / read from shard 0 */
int cmi_group_set_state(cmi_group_handle g, int which, const void *src, int64_t count, int dtype);
int cmi_group_get_state(cmi_group_handle g, int which, void *dst, int64_t count, int dtype);
int cmi_group_train_epoch(cmi_group_handle g, double lrate, double *loss_out);
/* local learning rate of every shard = lrate x scale when the group has more than one shard (default 1).  The mean merge divides an
 * item row's move by the shard count, so at equal rate a sharded run needs more epochs for the same training RMSE (1.2x / 1.4x / 1.6x
 * at 2 / 4 / 8 shards); scale = sqrt(n_shards) recovers it to <= 1.2x -- what the Java / C++ hosts set */
int cmi_group_set_lr_scale(cmi_group_handle g, double scale);
int cmi_group_train(cmi_group_handle g, int num_iters, double init_lrate, double max_lrate, int bold_driver, double decay, int early_stop,
                    double *losses, double *lrates, int *iters_run, double *final_lrate);
int cmi_group_train_from(cmi_group_handle g, int first_iter, double prev_loss, int num_iters, double init_lrate, double max_lrate,
                         int bold_driver, double decay, int early_stop, double *losses, double *lrates, int *iters_run, double *final_lrate);
/* test tuples are routed to the shard that owns their user; the error sums are merged exactly */
int cmi_group_eval_ratings(cmi_group_handle g, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, const double *r,
                           double min_rate, double max_rate, double *out, int64_t *count);
int cmi_group_predict_batch(cmi_group_handle g, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, int bound, double lo,
                            double hi, double *out);
/* info[0..1] = [first, last) user of the shard, info[2] = its tuples, info[3] = its device, info[4] = exchange in use (0 none: one
 * shard, 1 RCCL, 2 in-process), info[5] = elements of the exchanged bucket */
int cmi_group_shard_info(cmi_group_handle g, int shard, int64_t info[6]);
/* measurement (no reference counterpart; the reference's fold threads are timed by Recommender.execute()'s wall clock,
 * Recommender.java:284-297): HIP-event times of the most recent cmi_group_train_epoch, one float per shard -- compute_ms = the local
 * epoch's launches, exchange_ms = pack .. apply on the shard's stream (the collectives or the in-process sums, including the wait for
 * the slowest shard; 0 for a group of one) */
int cmi_group_last_times(cmi_group_handle g, float *compute_ms, float *exchange_ms);
/* Which exchange the group runs and why, as text (valid after cmi_group_set_ratings): RCCL -- after a pre-flight that ran one small
 * exchange of integer values through RCCL and through the in-process path and found both bit-identical to the host's sums -- or the
 * in-process exchange (peer copies): because shards share a device, or as the FALLBACK when ncclCommInitAll or the pre-flight failed
 * on this node (the reason is in the text and on stderr; training works either way).  No reference counterpart (CARSKit.java:395-412
 * has threads, not devices). */
const char *cmi_group_exchange_path(cmi_group_handle g);
/* the shard's instance (owned by the group), e.g. for cmi_schedule_info / cmi_last_epoch_ms */
int cmi_group_member(cmi_group_handle g, int shard, cmi_handle *out);
/* `--early-stop MAE|RMSE` for a sharded recommender (IterativeRecommender.java:149-161: isConverged() scores the test set after every
 * epoch): the test tuples are routed ONCE to the shard that owns their user and stay on its device (cmi_set_eval_ratings per shard);
 * cmi_group_eval_resident adds the shards' sums in shard order -> out = {MAE, RMSE, NMAE, rMAE, rRMSE} as cmi_eval_resident */
int cmi_group_set_eval_ratings(cmi_group_handle g, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, const double *r);
int cmi_group_eval_resident(cmi_group_handle g, double min_rate, double max_rate, double out[5], int64_t *count);

/* ---- the same exchange for a one-process-per-GPU job (carskit_amd/dist.py, bench.py --gpus N under torch.distributed.run) ------------
 * ONE implementation serves both hosts: cmi_group_* (one process, ncclCommInitAll) and cmi_comm_* (ncclCommInitRank) issue the same
 * three collectives from the same function (group_api.cpp, exchange_collective).  The host ranks share CMI_COMM_ID_BYTES bytes once
 * (rank 0 calls cmi_comm_unique_id, the host's own rendezvous hands the bytes to the other ranks), every rank calls cmi_comm_init
 * on its handle (after cmi_set_ratings / cmi_set_state: it snapshots the item-side containers), then per epoch
 *   cmi_comm_train_epoch(h, lrate, scale, &loss) = local epoch -> pack -> reduce-scatter + all-gather -> apply(scale) -> loss all-reduce
 * with scale = 1/world (the mean of the ranks' moves); `loss` is the GLOBAL loss, the epoch's one host
 * synchronisation.  cmi_comm_exchange alone is the exchange without the epoch (everything enqueued on cmi_stream()). */
#define CMI_COMM_ID_BYTES 128
int cmi_comm_unique_id(void *id /* CMI_COMM_ID_BYTES */);
int cmi_comm_init(cmi_handle h, const void *id, int rank, int world);
int cmi_comm_exchange(cmi_handle h, double scale);
int cmi_comm_train_epoch(cmi_handle h, double lrate, double scale, double *global_loss);
/* HIP-event time of the most recent cmi_comm_exchange on the handle's stream (pack .. loss all-reduce, including the wait for the
 * slowest rank); cmi_last_epoch_ms is the local epoch beside it */
int cmi_comm_last_exchange_ms(cmi_handle h, float *ms);

/* ---- FM: src/carskit/alg/cars/adaptation/dependent/FM.java (ALS / coordinate-descent sweep, not SGD) ---------
 * Separate handle type: the state is (w0, w[p], V[p x k]) with p = numUsers+numItems+numConditions
 * (FM.java:57-74) plus the per-rating errors[] and Q[][] of buildModel() (FM.java:117-146).  fp64 on the
 * device (the reference's precision).  The reference's quirks are kept: the context feature of a rating is
 * index numUsers+numItems+c with c the CONTEXT-COMBINATION id and value 1/numContextDims, present only if
 * c < numConditions (FM.java:81-86); denominators add the regulariser once per rating (FM.java:181,201); the
 * error/Q update of a factor uses x_il (FM.java:209-210).  errors[] and Q[][] are not stored: an error is
 * err0 + the running delta sums of its three coordinates (fm_kernels.hip), and a coordinate's sums are added per
 * slice of its support (not the Java's sequential order): model within ~1e-8 relative of the reference
 * arithmetic, RMSE within 1e-9.  `loss` (FM.java:218) is never
 * read by the reference and is not computed. */
typedef struct cmi_fm_instance *cmi_fm_handle;

/* new FM(train, test, fold) + initModel() allocation (FM.java:49-74).  Default (flags 0): every coordinate's sums are added in an order
 * the data layout alone decides (LDS parking + a fixed walk) -- two runs on identical inputs give bit-identical models, as two runs of
 * the reference's sweep do (FM.java:148-218), and the split phases (reduce / apply) equal the fused sweep bit for bit.
 * CMI_FM_FLAG_RELAXED_SUMS (or env CMI_FM_RELAXED_SUMS=1): the records' products are added with LDS atomics as they are evaluated --
 * ~11 % less time per sweep, same sums to the last few bits, but the order of the fp64 additions (so the last bits of the model, ~1e-12
 * relative) varies from run to run.  Both forms are within the 1e-8 the FM path promises against the reference arithmetic.
 * CMI_FM_FLAG_DETERMINISTIC: the round-5 name of what is now the default; accepted, no effect. */
#define CMI_FM_FLAG_DETERMINISTIC 0x1u
#define CMI_FM_FLAG_RELAXED_SUMS 0x2u
int cmi_fm_create(int k, int n_users, int n_items, int n_conds, int n_ctx_dims, int device, unsigned flags,
                  cmi_fm_handle *out);
int cmi_fm_destroy(cmi_fm_handle h);
const char *cmi_fm_last_error(cmi_fm_handle h);
/* -lw / -lf of the `FM=` line (FM.java:53-54; Java floats promoted) and `size` = trainMatrix.size() over ALL
 * ranks (FM.java:64; <= 0 means the local tuple count) */
int cmi_fm_set_hparams(cmi_fm_handle h, double regLw, double regLf, int64_t global_size);
/* training tuples as `trainMatrix.iterator()` yields them (FM.java:118-127); ctx = context-combination id */
int cmi_fm_set_ratings(cmi_fm_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx,
                       const double *r);
/* w0, w (p), V (p x k row-major): injected initial model / trained model (FM.java:65-70) */
int cmi_fm_set_model(cmi_fm_handle h, double w0, const double *w, const double *V);
int cmi_fm_get_model(cmi_fm_handle h, double *w0, double *w, double *V);
/* the pre-pass of buildModel(): errors[] and Q[][] from the current model (FM.java:117-146) */
int cmi_fm_init(cmi_fm_handle h);
/* one iteration of the `for (iter ...)` loop (FM.java:148-218) on one GPU */
int cmi_fm_sweep(cmi_fm_handle h);
/* whole buildModel(): cmi_fm_init + num_iters sweeps (the reference has no early stop for FM) */
int cmi_fm_train(cmi_fm_handle h, int num_iters);
/* FM.predict (FM.java:93-113) (+ bounding, Recommender.java:306-317) */
int cmi_fm_predict_batch(cmi_fm_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx,
                         int bound, double lo, double hi, double *out);
int cmi_fm_synchronize(cmi_fm_handle h);
/* the HIP stream (hipStream_t) all of this handle's work is enqueued on (a multi-GPU host orders its collectives on it) */
int cmi_fm_stream(cmi_fm_handle h, void **stream);
/* Recommender.evalRankings (Recommender.java:668-964) with FM.predict (FM.java:93-113) as the scorer: arguments, outputs
 * and semantics exactly as cmi_eval_rankings above */
int cmi_fm_eval_rankings(cmi_fm_handle h, int64_t n_train, const int32_t *tu, const int32_t *tj, const int32_t *tctx,
                         const double *tr, int64_t n_test, const int32_t *su, const int32_t *sj, const int32_t *sctx,
                         const double *sr, double bin_thold, int num_recs, int num_ignore, int strategy, double out[21],
                         int64_t *n_queries, int32_t *q_user, int32_t *q_ctx, int32_t *q_count, int32_t *top_items,
                         double *top_scores);
/* multi-GPU plumbing (no reference counterpart): a sweep is cmi_fm_num_phases() phases (0: w0; 1-3: w of the
 * user / item / context-feature field; 4+3f+field: column f of V).  phase_reduce leaves the local partial sums
 * [num(count/2) | den(count/2)] in a device buffer the host may all-reduce (ratings sharded across ranks),
 * phase_apply consumes it.  reduce+apply over all phases == cmi_fm_sweep. */
int cmi_fm_num_phases(cmi_fm_handle h);
int cmi_fm_phase_reduce(cmi_fm_handle h, int phase);
int cmi_fm_phase_buffer(cmi_fm_handle h, int phase, void **dev_ptr, int64_t *count);
int cmi_fm_phase_apply(cmi_fm_handle h, int phase);
/* reduce + apply of one phase fused (no exchange point), for phases whose coordinates are local to the rank */
int cmi_fm_phase_run(cmi_fm_handle h, int phase);
/* measurement (no reference counterpart): the layout cmi_fm_set_ratings built -- out[0..1] slices of the user / item order,
 * [2..4] records of the three orders, [5..6] chunks, [7] HBM bytes one factor has to move, [8..9] of that the reduce launch of
 * the user / item field, [10] slice entries, [11] p -- and the HIP-event duration of one phase's reduce kernel (it writes only
 * scratch, the model is untouched) */
/* ratings sharded by user over one process per GPU: the sweep with its per-phase exchange (all-reduce of [num | den] for the w0 /
 * item / context phases; user phases are rank-local) issued by the library on the instance's stream.  `id`: CMI_COMM_ID_BYTES bytes of
 * cmi_comm_unique_id() from rank 0; cmi_fm_set_hparams' global_size = the ratings of ALL ranks. */
int cmi_fm_comm_init(cmi_fm_handle h, const void *id, int rank, int world);
int cmi_fm_comm_sweep(cmi_fm_handle h);
int cmi_fm_layout(cmi_fm_handle h, int64_t out[12]);
int cmi_fm_time_reduce(cmi_fm_handle h, int phase, int reps, double *avg_ms);

/* ---- data side of the path (host-only, no GPU): DataDAO id-mapper and the compact->binary rewrite -------------
 * Integer / string work that must be BIT-EXACT with the reference (north_star: "integer id mapping bit-exact"). */
typedef struct cmi_dao *cmi_dao_handle;

/* DataDAO.readData (src/carskit/data/processor/DataDAO.java:166-354) for a binary-format rating file:
 * header tokens >= 3 are conditions (`dim:cond`), data lines `user,item,rating,0/1...`; inner ids are assigned
 * in first-seen order (users, items, "u,i" pairs, context keys = active condition indices joined by ','),
 * duplicates of a (pair, context) cell: last line wins; rating scale = sorted distinct values. */
int cmi_dao_read(const char *path, cmi_dao_handle *out);
/* `test-set` evaluation: the test DAO is built over the TRAINING DAO's id maps and extends them with unseen
 * users/items/contexts (src/carskit/main/CARSKit.java:335-340, DataDAO.java:119-143).  The result's counts and raw-id
 * tables are those of the union (what rateDao.numUsers() etc. return afterwards); its matrix is the test matrix. */
int cmi_dao_read_shared(const char *path, cmi_dao_handle train, cmi_dao_handle *out);
int cmi_dao_destroy(cmi_dao_handle h);
const char *cmi_dao_last_error(cmi_dao_handle h);
/* out: numUsers, numItems, numUserItems, numContexts, numConditions, numContextDims, numRatings (lines), matrix entries */
int cmi_dao_counts(cmi_dao_handle h, int64_t out[8]);
/* the (user-item x context) rating matrix in MatrixIterator (CRS) order */
int cmi_dao_matrix(cmi_dao_handle h, int32_t *ui, int32_t *ctx, double *r);
/* getUserIdFromUI / getItemIdFromUI (DataDAO.java:1038-1046) as arrays over pair ids */
int cmi_dao_ui_maps(cmi_dao_handle h, int32_t *ui_user, int32_t *ui_item);
/* getContextConditionsList as CSR (ctx_ptr has numContexts+1 entries; cmi_dao_ctx_nnz gives len(ctx_conds)) */
int64_t cmi_dao_ctx_nnz(cmi_dao_handle h);
int cmi_dao_ctx_table(cmi_dao_handle h, int32_t *ctx_ptr, int32_t *ctx_conds);
/* condDimensionMap (numConditions entries) and EmptyContextConditions (conditions whose token ends with ":na") */
int cmi_dao_cond_info(cmi_dao_handle h, int32_t *cond_dim, int32_t *empty_conds, int32_t *n_empty);
int cmi_dao_rating_scale(cmi_dao_handle h, double *out, int32_t cap, int32_t *n);
/* raw key of an inner id; kind: 0 user, 1 item, 2 condition, 3 context key, 4 dimension, 5 "u,i" pair key */
const char *cmi_dao_raw_id(cmi_dao_handle h, int kind, int32_t idx);

/* Iteration order of a default java.util.HashMap<String,?> (JDK 8+) after inserting n DISTINCT keys in the given
 * order: positions[i] = index of the i-th key `for (k : map.keySet())` visits.  *treeified = 1 if a bin reached
 * the treeify threshold (order then not guaranteed).  Used by the transformer below. */
int cmi_java_hashmap_order(int64_t n, const char *const *keys, int64_t *positions, int *treeified);
/* DataTransformer.TransformationFromCompactToBinary + PublishNewRatingFiles + getHeader
 * (src/carskit/data/processor/DataTransformer.java:231-259,266-329,155-163): rewrites a compact-format file
 * (user,item,rating,dim1,dim2,...) as the binary-format train.csv, rows in the reference's HashMap order. */
int cmi_transform_compact_to_binary(const char *in_path, const char *out_path, int *treeified);
/* CARSKit.validateDataFormat (src/carskit/main/CARSKit.java:179-215): 1 binary, 2 loose, 3 compact; 0 where the reference throws
 * (no data line, a one-column header, a data line shorter than the header, a value Integer.valueOf refuses under a dim:cond column) */
int cmi_validate_data_format(const char *path);
/* DataTransformer.run() (DataTransformer.java:331-396) for binary / loose / compact input.  test_in == NULL: only the
 * training file (binary is copied).  With a test file, both are rewritten against the merged, SORTED condition set
 * of getConditions() (DataTransformer.java:57-92). */
int cmi_transform(const char *train_in, const char *train_out, const char *test_in, const char *test_out, int *treeified);

/* ---- host-only integer preprocessing (runs without a GPU) ------------------------------------- */

/* The dependency-level schedule the default mode executes (carskit_amd/csrc/level_schedule.cpp):
 * level(t) = 1 + max(level of the previous tuple with the same user, ... same item).  It replaces the
 * reference's implicit "one tuple after another" order (librec MatrixIterator, CAMF_CI.java:80) by the
 * weakest order that yields the identical result.  perm[n]: schedule position -> CRS tuple index;
 * level_off[0..*n_levels]: offsets of the levels in perm (level_cap = capacity of level_off, must be
 * >= *n_levels + 1; pass level_off = NULL to only count).  order: 0 keep CRS order inside a level,
 * 1 sort by item id, 2 sort by user id (tuples of one level commute, so this is free). */
int cmi_level_schedule(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int order,
                       int32_t *perm, int64_t *level_off, int64_t level_cap, int64_t *n_levels);

/* The hub-chain form of the level schedule (level_schedule.cpp, build_chain_schedule; the default execution order).  hub: 1 chain
 * along items, 0 along users, -1 pick the one with fewer units, -2 / -3 users / items unless that costs more than 1.3x the units of the
 * other side (what cmi_set_ratings uses for CAMF_CU / CAMF_CI: the side with the context-bias rows) (*hub_used reports the choice).  perm[n]: stream position -> CRS tuple;
 * unit_off[*n_units+1]: offsets of the units in perm; level_off[*n_levels+1]: offsets of the levels in unit_off (unit indices).
 * Pass perm = NULL to only count (*n_units, *n_levels). */
int cmi_chain_schedule(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int hub, int max_chain,
                       int32_t *perm, int32_t *unit_off, int64_t unit_cap, int64_t *level_off, int64_t level_cap,
                       int64_t *n_units, int64_t *n_levels, int *hub_used);
/* the same schedule built on `device` (sched_device.hip: a lane per hub row walks the row's CRS chain, the spoke rows hand over through
 * atomic words; sorts and scans around it) -- what cmi_set_ratings uses from 2 M tuples on; element for element the host's result, the
 * order of librec's MatrixIterator restated, not changed.  CMI_E_UNSUPPORTED: not built (the caller uses cmi_chain_schedule). */
int cmi_chain_schedule_device(int device, int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int hub, int max_chain,
                              int32_t *perm, int32_t *unit_off, int64_t unit_cap, int64_t *level_off, int64_t level_cap,
                              int64_t *n_units, int64_t *n_levels, int *hub_used);

/* host-only views of two schedule post-passes (tests).  cmi_narrow_runs: run_len[l] > 0 = a run of that many consecutive
 * levels with <= max_tuples tuples each starts at level l and is walked by ONE launch (the library uses 256 / 16), -1 = inside
 * a run, 0 = own launch; *n_launches = launches per epoch.  cmi_conflict_free_blocks: CAMF_C's CRS blocks -- maximal runs of
 * consecutive tuples sharing no user and no item, cut at max_block (the library uses 64); off may be NULL to query *n_blocks. */
int cmi_narrow_runs(int64_t n_levels, const int64_t *level_off, int64_t max_tuples, int64_t min_levels, int32_t *run_len,
                    int64_t *n_launches);
int cmi_conflict_free_blocks(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int32_t max_block,
                             int32_t *off, int64_t off_cap, int64_t *n_blocks);

/* The owner form behind CMI_FLAG_SCHED_OWNER (level_schedule.cpp, build_owner_schedule): like the level schedules it replaces the
 * reference's implicit "one tuple after another" order (librec MatrixIterator in `for (MatrixEntry me : trainMatrix)`,
 * CAMF_CI.java:80) by an explicit order-exact one -- every row of the heavy side is walked by one owner in CRS order, the other side's
 * rows wait for their update count.  hub: 1 items are owned, 0 users, -1 the side
 * with the larger maximum degree (*hub_used reports it).  perm[n]: list position -> CRS tuple (an owner's tuples contiguous, in CRS
 * order); own_off[n_owners+1]; want[n]: updates of the tuple's spoke row that precede it; flags[n]: bit 0 hub row taken over in
 * registers from the previous list entry, bit 1 hub row re-read late (written < depth+1 entries back), bit 2 hub row stored, bit 3
 * spoke row taken over in registers, bit 4 spoke record stored. */
int cmi_owner_schedule(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int hub, int n_owners, int depth,
                       int32_t *perm, int64_t *own_off, uint32_t *want, uint32_t *flags, int *hub_used);

#ifdef __cplusplus
}
#endif
#endif

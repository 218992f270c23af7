// rank_host.hpp -- model-agnostic core of the ranking evaluation (Recommender.evalRankings,
// src/carskit/generic/Recommender.java:668-964), shared by cmi_eval_rankings (MF family) and cmi_fm_eval_rankings (FM):
// host bookkeeping (candidates, queries, exclusions), the device scoring driver, the metric formulas.  Internal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <chrono>
#include <functional>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace cmi {

struct RankPlan {
    std::vector<int32_t> cand;            // candidate position -> item id (HashSet<Integer> order, minus ignored)
    std::vector<int32_t> qu, qc;          // query -> user, context
    std::vector<int32_t> gu, qg, gq0;     // query groups = runs of one user: group -> user; query -> group; group -> first query (+ end)
    std::vector<int32_t> truth_items;     // per query: its positive test items that are candidates (sorted)
    std::vector<int64_t> truth_ptr;
    std::vector<int32_t> excl_idx;        // per query: candidate positions rated by the user in that context (training)
    std::vector<int64_t> excl_ptr;
};

struct RankTuples {
    int64_t n;
    const int32_t *u, *j, *ctx;
    const double *r; // may be null for training tuples (all non-zero)
};

// ids must already be range-checked by the caller
void rank_build_plan(int n_users, int n_items, const RankTuples &train, const RankTuples &test, double bin_thold,
                     int num_ignore, RankPlan &plan);

// operand builders of a model: B[c] rows for candidates, A[q] rows + per-row constants for a batch of queries
template <typename T>
struct RankOperands {
    int k_logical; // un-padded operand length
    std::function<hipError_t(T *dB, const int32_t *dcand, int nc, int kp, hipStream_t)> build_items;
    std::function<hipError_t(T *dA, T *drc, const int32_t *dqu, const int32_t *dqc, int n, int kp, hipStream_t)> build_queries;
};

// Device and pinned-host buffers of an evaluation, owned by the handle and reused (grow-only) by the next one: allocating and
// freeing ~1 GiB per call costs milliseconds and a device synchronisation each.
struct RankWorkspace {
    struct Buf {
        void *p = nullptr;
        size_t cap = 0;
    };
    Buf dA, dB, dS, drc, dcand, dqu, dqc, dexptr, dexcl, dtop, dscore, dcount; // device
    Buf dB2, dA2, dS2, dqg, dqd, dgu, ddc, dscr;                                // device, split form (rank_run_device_split)
    Buf dcolc;                                                                  // split form: itemBias of the candidates (the S1 contraction adds it at the end)
    Buf dSb, dAb;                                                               // split form: the second slab / operand buffer (batch b + 1 is contracted while batch b is selected)
    Buf dM1, dM1b, dM2;                                                         // split form: per-row maxima of S1 (per slab) / S2 over tiles of 64 candidates (the selection's tile pruning)
    hipStream_t sel_stream = nullptr;                                           // split form: the selection's stream
    hipStream_t gemm_stream = nullptr;                                          // experiment builds (CMI_RANK_SEL_CUS): the contraction on the compute units the selection's masked stream leaves
    hipEvent_t ev_gs = nullptr;
    std::vector<hipEvent_t> evgemm, evsel;                                           // per batch: contraction done (main stream), selection done (selection stream)
    Buf h_top, h_score, h_count;                                                // pinned host: the lists as they come back
    RankPlan plan;                                                              // the last evaluation's plan (capacity is reused)
    std::vector<int32_t> v_dctx, v_qd;                                          // split form: context index arrays (capacity is reused)
    bool ctx_ready = false;                                                     // v_dctx / v_qd hold this evaluation's contexts (rank_split_usable)
    // Repeated evaluations of the same (train, test) tuples (`--early-stop` on a ranking measure evaluates after every epoch,
    // IterativeRecommender.java:149-161): the plan depends on the tuples alone, so it is kept, keyed by their sizes and a 64-bit
    // content hash, and so are the index arrays it put on the device.
    struct PlanKey {
        int64_t n_train = -1, n_test = -1, n_users = 0, n_items = 0;
        double bin_thold = 0;
        int num_ignore = 0;
        uint64_t hash = 0;
        bool operator==(const PlanKey &o) const {
            return n_train == o.n_train && n_test == o.n_test && n_users == o.n_users && n_items == o.n_items && bin_thold == o.bin_thold &&
                   num_ignore == o.num_ignore && hash == o.hash;
        }
    } plan_key;
    bool plan_valid = false;      // `plan` belongs to plan_key
    const void *resident[9] = {}; // split form: the device buffers that hold plan_key's index arrays (null: not uploaded)
    struct HostVals { // per-query measures, uninitialised and grow-only (38 MB for 270 K queries: not re-faulted per call)
        std::unique_ptr<double[]> p;
        size_t cap = 0;
        double *need(size_t n) {
            if (n > cap) {
                p.reset(new double[n + n / 8]);
                cap = n + n / 8;
            }
            return p.get();
        }
    } vals, umeans; // umeans: per-user means of the ucu strategy (rank_fold_users)
    hipEvent_t ev0 = nullptr, ev1 = nullptr; // around the device loop (timing)
    std::vector<hipEvent_t> evb;              // one per batch: its lists have arrived on the host
    std::vector<hipEvent_t> evk;              // three per batch of the split form: before / after the contraction, after the selection
    double kernel_ms[2] = {0, 0};             // last evaluation: contraction, selection (summed over the batches)
    hipError_t kernel_event(size_t i, hipStream_t stream);
    // host wall clock of the last evaluation, ms: [0] plan, [1] setup (buffers, uploads, item operands), [2] scoring loop incl. the
    // overlapped per-batch measures, [3] tail (last batch's measures + the averages), [4] total
    double host_ms[5] = {0, 0, 0, 0, 0};
    hipError_t need(Buf &b, size_t bytes, bool pinned = false);
    void release(); // the caller has made the owning device current
    hipError_t batch_event(size_t b, hipStream_t stream);
    // waits for the batches in order and hands each to on_batch while the device works on the later ones; fills host_ms[2], [3]
    hipError_t consume_batches(hipError_t e, const std::vector<std::pair<int64_t, int64_t>> &batches,
                               const std::function<void(int64_t, int64_t)> &on_batch, std::chrono::steady_clock::time_point t_loop);
};

// on_batch(q0, q1): the lists of queries [q0, q1) have arrived in ws.h_top / h_score / h_count (absolute query indexing); called on
// the host while the device scores the next batch
template <typename T>
hipError_t rank_run_device(hipStream_t stream, RankWorkspace &ws, const RankPlan &plan, const RankOperands<T> &ops, double thold, int topn,
                           const std::function<void(int64_t, int64_t)> &on_batch, float *ms, double *flops);

// The split form for the MF family in fp32 (rank_kernels.hip): S1 per distinct query user, S2 per distinct context, added by the selection.
struct RankSplitArgs;
hipError_t rank_run_device_split(hipStream_t stream, RankWorkspace &ws, const RankPlan &plan, RankSplitArgs base, double thold, int topn,
                                 const std::function<void(int64_t, int64_t)> &on_batch, float *ms, double *flops);
// whether the split form applies: S2 must stay small (distinct contexts x candidates) and the lists must fit the register selection
bool rank_split_usable(const RankPlan &plan, int topn, RankWorkspace &ws);

// the 18 measures of the lists of queries [q0, q1) -> vals[q * 18 + m] (threads over ranges of queries); optional per-query outputs
void rank_measures_range(const RankPlan &plan, int num_recs, const int32_t *top_idx, const double *top_score, const int32_t *top_count,
                         int64_t q0, int64_t q1, double *vals, int32_t *q_user, int32_t *q_ctx, int32_t *q_count, int32_t *top_items,
                         double *top_scores);
// averaged per strategy, in query order (Recommender.java:850-960).  ucu: rank_fold_users writes every user's mean over its contexts
// into `umeans` (18 doubles per user, user order) and may run batch by batch behind the device for the users a batch completes;
// rank_average folds what is left and sums over the users (umeans == nullptr: it allocates its own and folds everything).
struct RankFolded {
    int64_t q = 0, u = 0; // first query not folded yet; users folded so far
    // the serial sum over users (ucu) / queries (uc), advanced batch by batch behind the device in the order rank_average would take
    int64_t summed = 0;   // users (ucu) or queries (uc) already in s / c
    double s[18] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int64_t c[18] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};
// uc: add the queries [f.summed, q_to) to the running sums (query order)
void rank_sum_queries(const int32_t *top_count, const double *vals, RankFolded &f, int64_t q_to);
void rank_fold_users(const RankPlan &plan, const int32_t *top_count, const double *vals, double *umeans, RankFolded &f, int64_t q_to, bool last);
void rank_average(const RankPlan &plan, int strategy, const int32_t *top_count, const double *vals, double *umeans, RankFolded f,
                  double *out /*[21]*/);

} // namespace cmi

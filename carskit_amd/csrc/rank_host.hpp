// rank_host.hpp -- model-agnostic core of the ranking evaluation (Recommender.evalRankings,
// src/carskit/generic/Recommender.java:668-964), shared by cmi_eval_rankings (MF family) and cmi_fm_eval_rankings (FM):
// host bookkeeping (candidates, queries, exclusions), the device scoring driver, the metric formulas.  Internal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

namespace cmi {

struct RankPlan {
    std::vector<int32_t> cand;            // candidate position -> item id (HashSet<Integer> order, minus ignored)
    std::vector<int32_t> qu, qc;          // query -> user, context
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

template <typename T>
hipError_t rank_run_device(hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, const RankPlan &plan, const RankOperands<T> &ops,
                           double thold, int topn, std::vector<int32_t> &top_idx, std::vector<double> &top_score,
                           std::vector<int32_t> &top_count, float *ms, double *flops);

void rank_metrics(const RankPlan &plan, int strategy, int num_recs, const std::vector<int32_t> &top_idx,
                  const std::vector<double> &top_score, const std::vector<int32_t> &top_count, double *out /*[21]*/,
                  int32_t *q_user, int32_t *q_ctx, int32_t *q_count, int32_t *top_items, double *top_scores);

} // namespace cmi

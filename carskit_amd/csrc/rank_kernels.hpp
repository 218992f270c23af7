// rank_kernels.hpp -- device side of cmi_eval_rankings (see rank_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmi {

template <typename T>
struct RankItemsArgs {
    const T *Q, *itemBias, *icBias; // itemBias / icBias may be null (model does not own them)
    const int32_t *cand;            // candidate position -> item id
    T *B;                           // [n_cand][kp]
    int n_cand, k, kp, n_conds;
};

template <typename T>
struct RankQueryArgs {
    const T *P, *userBias, *ucBias, *condBias; // optional containers null
    const int32_t *qu, *qc;                    // query -> user id, context id (this batch)
    const int32_t *ctx_ptr, *ctx_conds;        // context id -> condition ids (null for the 2-D models)
    T *A;                                      // [nq][kp]
    T *row_const;                              // [nq]
    double gm;
    int k, kp, n_conds, icBias_used;
};

template <typename T>
hipError_t rank_launch_build_items(const RankItemsArgs<T> &a, hipStream_t s);
template <typename T>
hipError_t rank_launch_build_queries(const RankQueryArgs<T> &a, int nq, hipStream_t s);
// S = A.B^T + row_const; mask the (q_base+q)-th exclusion list; top-N per row into out_*[(q_base+q)*topn + n]
template <typename T>
hipError_t rank_launch_score(const T *A, const T *B, const T *row_const, T *S, int nq, int nc, int kp,
                             const int64_t *excl_ptr, const int32_t *excl_idx, int q_base, double thold, int topn,
                             int32_t *out_idx, double *out_score, int32_t *out_count, hipStream_t s);

} // namespace cmi

// rank_kernels.hpp -- device side of cmi_eval_rankings (see rank_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmi {

// bytes the score slabs (S1, S2) are allocated beyond their last row: the pruned selection loads whole tiles of 64 candidates, and the
// last tile of a row may reach up to 63 elements past it (into the next row, or past the slab's last row into this slack; never used)
constexpr size_t RANK_SLAB_SLACK = 256;

template <typename T>
struct RankItemsArgs {
    const T *Q, *itemBias, *icBias; // itemBias / icBias may be null (model does not own them)
    const int32_t *cand;            // candidate position -> item id
    T *B;                           // [n_cand][kp]
    int n_cand, k, kp, n_conds;
    T *bias_out = nullptr;          // not null: itemBias[cand] goes HERE (a per-candidate constant the contraction adds at the end) and not into column k
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

// FM (FM.java:93-113): score = w0 + w_u + w_j + xc*w_c + <Vu,Vj> + xc*<Vu,Vc> + xc*<Vj,Vc>  (context feature only if c < n_conds)
//   b_j = [V[item j] | w[item j]],  a_q = [V[u] + xc*V[c] | 1],  row constant = w0 + w_u + xc*w_c + xc*<Vu,Vc>
struct RankFmArgs {
    const double *w0, *w, *V; // V: p x k row-major, p = n_users + n_items + n_conds
    int k, kp, n_users, n_items, n_conds;
    double xc;
};
hipError_t rank_launch_fm_items(const RankFmArgs &a, const int32_t *cand, int nc, double *B, hipStream_t s);
hipError_t rank_launch_fm_queries(const RankFmArgs &a, const int32_t *qu, const int32_t *qc, int nq, double *A, double *row_const,
                                  hipStream_t s);

template <typename T>
hipError_t rank_launch_build_items(const RankItemsArgs<T> &a, hipStream_t s);
template <typename T>
hipError_t rank_launch_build_queries(const RankQueryArgs<T> &a, int nq, hipStream_t s);
// S = A.B^T + row_const; mask the (q_base+q)-th exclusion list; top-N per row into out_*[(q_base+q)*topn + n]
template <typename T>
hipError_t rank_launch_score(const T *A, const T *B, const T *row_const, T *S, int nq, int nc, int kp,
                             const int64_t *excl_ptr, const int32_t *excl_idx, int q_base, double thold, int topn,
                             int32_t *out_idx, double *out_score, int32_t *out_count, hipStream_t s);


// the split form of the MF family's scores (rank_kernels.hip, "the split form"): fp32 state
struct RankSplitArgs {
    const float *P, *Q, *userBias, *itemBias, *ucBias, *icBias, *condBias; // optional containers null
    const int32_t *ctx_ptr, *ctx_conds;                                    // null for the 2-D models
    const int32_t *cand, *qu, *qc, *dctx;                                  // candidates; per query user / context; distinct contexts
    float *B1, *B2, *A2, *rc;                                              // operands built once per evaluation
    float *colc = nullptr;                                                 // [nc] itemBias of the candidates (added by the S1 contraction's epilogue)
    double gm;
    int k, kp1, kp2, n_conds, nc, nq, n_dctx;
};
hipError_t rank_launch_split_operands(const RankSplitArgs &a, hipStream_t s);
hipError_t rank_launch_split_users(const RankSplitArgs &a, const int32_t *d_group_user, int n, float *A1, float *scratch_rc, hipStream_t s);
hipError_t rank_launch_split_select(const float *S1, const float *S2, const RankSplitArgs &a, const int32_t *q_group, const int32_t *q_dctx, int g_base,
                                    int q0, int nq, const int64_t *excl_ptr, const int32_t *excl_idx, double thold, int topn, int32_t *out_idx,
                                    double *out_score, int32_t *out_count, hipStream_t s, const float *M1 = nullptr, const float *M2 = nullptr);
// S = (A.B^T [+ col_const]) + row_const (the contraction alone).  col_const[c] is added to the finished dot product first: the same value
// as one more column {1 | col_const[c]} at the end of the k-ordered chain (fma(1, b, acc) = acc + b), without the 16 padded columns it costs
// tile_max (may be null): [nq][(nc + 63) / 64] row maxima of S over tiles of 64 candidates, for the selection's tile pruning (M1 / M2 above)
template <typename T>
hipError_t rank_launch_gemm(const T *A, const T *B, const T *row_const, T *S, int nq, int nc, int kp, hipStream_t s, const T *col_const = nullptr,
                            T *tile_max = nullptr);

} // namespace cmi

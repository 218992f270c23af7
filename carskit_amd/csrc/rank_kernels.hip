// rank_kernels.hip -- scoring + top-N selection for the reference's ranking evaluation
// (Recommender.evalRankings, src/carskit/generic/Recommender.java:668-964): for every test (user, context) query,
// score ALL candidate items with predict(u, j, c), drop the items the user already rated in that context, keep the
// numRecs best.  The reference does this with one predict() call per (query, item): O(queries x items x k).
//
// Here the j-dependent part of every model's predict() is ONE dense contraction.  With the augmented vectors
//     a_q = [ P[u] | 1 | onehot(conditions of c) ]          b_j = [ Q[j] | itemBias[j] | icBias[j, :] ]
// <a_q, b_j> = <P[u],Q[j]> + itemBias[j] + sum_{cond in c} icBias[j, cond]; the remaining terms (globalMean, userBias[u],
// sum ucBias[u, cond], sum condBias[cond]) are constant along a query row and are added in the epilogue.
// So scoring is a (queries x K') x (K' x items) GEMM -- the one GEMM-shaped piece of this code base.  Round 1 runs it
// as an LDS-tiled VALU kernel in the state's own precision (fp32 or fp64): exact fp32/fp64 products and sums, no
// reduced-precision MFMA; the f32-input MFMA (v_mfma_f32_32x32x2_f32, same rate as VALU on gfx950 but less issue
// pressure) is the obvious next step for this kernel.
// Top-N: one wave64 per query row repeatedly extracts the row maximum (ties -> lowest candidate index, which
// reproduces the reference's stable descending sort over its candidate iteration order).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rank_kernels.hpp"

namespace cmi {

// ---- gather the augmented operands ---------------------------------------------------------------------------

template <typename T>
__global__ void rank_build_items(RankItemsArgs<T> a) { // one block per candidate item, threads over K'
    const int c = blockIdx.x;
    const int j = a.cand[c];
    T *dst = a.B + (size_t)c * a.kp;
    for (int f = threadIdx.x; f < a.kp; f += blockDim.x) {
        T v = 0;
        if (f < a.k) v = a.Q[(size_t)j * a.k + f];
        else if (f == a.k) v = a.itemBias ? a.itemBias[j] : (T)0;
        else if (f < a.k + 1 + a.n_conds) v = a.icBias ? a.icBias[(size_t)j * a.n_conds + (f - a.k - 1)] : (T)0;
        dst[f] = v;
    }
}

template <typename T>
__global__ void rank_build_queries(RankQueryArgs<T> a) { // one block per query
    const int q = blockIdx.x;
    const int u = a.qu[q], c = a.qc[q];
    T *dst = a.A + (size_t)q * a.kp;
    for (int f = threadIdx.x; f < a.kp; f += blockDim.x) {
        T v = 0;
        if (f < a.k) v = a.P[(size_t)u * a.k + f];
        else if (f == a.k) v = 1;
        dst[f] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // per-row constant: every predict() term that does not depend on the item
        T rc = (T)a.gm;
        if (a.userBias) rc += a.userBias[u];
        if (a.ctx_ptr) {
            for (int p = a.ctx_ptr[c]; p < a.ctx_ptr[c + 1]; ++p) {
                const int cond = a.ctx_conds[p];
                if (a.icBias_used) dst[a.k + 1 + cond] = 1; // one-hot of the context's conditions
                if (a.ucBias) rc += a.ucBias[(size_t)u * a.n_conds + cond];
                if (a.condBias) rc += a.condBias[cond];
            }
        }
        a.row_const[q] = rc;
    }
}

// ---- S[q][c] = <A[q,:], B[c,:]> + row_const[q]  (LDS-tiled, 64x64 tile, 4x4 per thread) -------------------------

template <typename T>
__global__ __launch_bounds__(256) void rank_gemm(const T *__restrict__ A, const T *__restrict__ B, const T *row_const,
                                                 T *__restrict__ S, int nq, int nc, int kp) {
    constexpr int TM = 64, TN = 64, TK = 16;
    __shared__ T sA[TK][TM + 1];
    __shared__ T sB[TK][TN + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int q0 = blockIdx.y * TM, c0 = blockIdx.x * TN;
    T acc[4][4] = {};
    for (int k0 = 0; k0 < kp; k0 += TK) {
        for (int e = threadIdx.x; e < TM * TK; e += 256) { // rows of A / B are contiguous in k
            const int r = e / TK, kk = e % TK;
            sA[kk][r] = (q0 + r < nq && k0 + kk < kp) ? A[(size_t)(q0 + r) * kp + k0 + kk] : (T)0;
            sB[kk][r] = (c0 + r < nc && k0 + kk < kp) ? B[(size_t)(c0 + r) * kp + k0 + kk] : (T)0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) {
            T av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = sA[kk][ty * 4 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[i] = sB[kk][tx * 4 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[i][jj] = fma(av[i], bv[jj], acc[i][jj]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = q0 + ty * 4 + i;
        if (q >= nq) continue;
        const T rc = row_const[q];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int c = c0 + tx * 4 + jj;
            if (c < nc) S[(size_t)q * nc + c] = acc[i][jj] + rc;
        }
    }
}

// ---- exclusions: items the user already rated in this context (never candidates) -----------------------------------

template <typename T>
__global__ void rank_mask(T *S, int nc, const int64_t *excl_ptr, const int32_t *excl_idx, int q_base, int nq) {
    const int q = blockIdx.x;
    if (q >= nq) return;
    for (int64_t p = excl_ptr[q_base + q] + threadIdx.x; p < excl_ptr[q_base + q + 1]; p += blockDim.x)
        S[(size_t)q * nc + excl_idx[p]] = -INFINITY;
}

// ---- top-N: one wave64 per query row ----------------------------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(256) void rank_topn(T *S, int nq, int nc, double thold, int topn, int32_t *out_idx,
                                                 double *out_score, int32_t *out_count, int q_base) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    T *row = S + (size_t)q * nc;
    int found = 0;
    for (int n = 0; n < topn; ++n) {
        T best = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = lane; c < nc; c += 64) {
            const T v = row[c];
            // candidates must satisfy `score > threshold` and not be NaN (Recommender.java:808-812)
            if ((double)v > thold && (v > best || (v == best && c < bi))) {
                best = v;
                bi = c;
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const T ov = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (ov > best || (ov == best && oi < bi)) {
                best = ov;
                bi = oi;
            }
        }
        if (bi == 0x7fffffff) break; // nothing left above the threshold
        if (lane == 0) {
            out_idx[(size_t)(q_base + q) * topn + n] = bi;
            out_score[(size_t)(q_base + q) * topn + n] = (double)best;
            row[bi] = -INFINITY;
        }
        ++found;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
    if (lane == 0) out_count[q_base + q] = found;
}

// ---- launchers -------------------------------------------------------------------------------------------------------

template <typename T>
hipError_t rank_launch_build_items(const RankItemsArgs<T> &a, hipStream_t s) {
    if (a.n_cand <= 0) return hipSuccess;
    hipLaunchKernelGGL(rank_build_items<T>, dim3(a.n_cand), dim3(128), 0, s, a);
    return hipGetLastError();
}
template <typename T>
hipError_t rank_launch_build_queries(const RankQueryArgs<T> &a, int nq, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(rank_build_queries<T>, dim3(nq), dim3(128), 0, s, a);
    return hipGetLastError();
}
template <typename T>
hipError_t rank_launch_score(const T *A, const T *B, const T *row_const, T *S, int nq, int nc, int kp,
                             const int64_t *excl_ptr, const int32_t *excl_idx, int q_base, double thold, int topn,
                             int32_t *out_idx, double *out_score, int32_t *out_count, hipStream_t s) {
    if (nq <= 0 || nc <= 0) return hipSuccess;
    hipLaunchKernelGGL(rank_gemm<T>, dim3((nc + 63) / 64, (nq + 63) / 64), dim3(256), 0, s, A, B, row_const, S, nq, nc, kp);
    hipLaunchKernelGGL(rank_mask<T>, dim3(nq), dim3(64), 0, s, S, nc, excl_ptr, excl_idx, q_base, nq);
    hipLaunchKernelGGL(rank_topn<T>, dim3((nq + 3) / 4), dim3(256), 0, s, S, nq, nc, thold, topn, out_idx, out_score,
                       out_count, q_base);
    return hipGetLastError();
}

#define CMI_INST(T)                                                                                                    \
    template hipError_t rank_launch_build_items<T>(const RankItemsArgs<T> &, hipStream_t);                             \
    template hipError_t rank_launch_build_queries<T>(const RankQueryArgs<T> &, int, hipStream_t);                      \
    template hipError_t rank_launch_score<T>(const T *, const T *, const T *, T *, int, int, int, const int64_t *,     \
                                             const int32_t *, int, double, int, int32_t *, double *, int32_t *, hipStream_t);
CMI_INST(float)
CMI_INST(double)
#undef CMI_INST

} // namespace cmi

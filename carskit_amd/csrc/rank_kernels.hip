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
#include "env_knobs.hpp"
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "rank_kernels.hpp"

namespace cmi {

// ---- gather the augmented operands ---------------------------------------------------------------------------

template <typename T>
__global__ void rank_build_items(RankItemsArgs<T> a) { // one block per candidate item, threads over K'
    const int c = blockIdx.x;
    const int j = a.cand[c];
    T *dst = a.B + (size_t)c * a.kp; // kp = padded row length; columns past k + 1 + n_conds stay zero
    for (int f = threadIdx.x; f < a.kp; f += blockDim.x) {
        T v = 0;
        if (f < a.k) v = a.Q[(size_t)j * a.k + f];
        else if (f == a.k) v = a.itemBias && !a.bias_out ? a.itemBias[j] : (T)0;
        else if (f < a.k + 1 + a.n_conds) v = a.icBias ? a.icBias[(size_t)j * a.n_conds + (f - a.k - 1)] : (T)0;
        dst[f] = v;
    }
    if (a.bias_out && threadIdx.x == 0) a.bias_out[c] = a.itemBias ? a.itemBias[j] : (T)0;
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

// ---- FM operands ------------------------------------------------------------------------------------------------------

__global__ void rank_fm_items(RankFmArgs a, const int32_t *cand, double *B) { // one block per candidate
    const int64_t l = (int64_t)a.n_users + cand[blockIdx.x];
    double *dst = B + (size_t)blockIdx.x * a.kp;
    for (int f = threadIdx.x; f < a.kp; f += blockDim.x) dst[f] = f < a.k ? a.V[(size_t)l * a.k + f] : (f == a.k ? a.w[l] : 0.0);
}

__global__ void rank_fm_queries(RankFmArgs a, const int32_t *qu, const int32_t *qc, double *A, double *row_const) {
    __shared__ double part[64];
    const int u = qu[blockIdx.x], c = qc[blockIdx.x];
    const bool has_c = c < a.n_conds; // the reference's index quirk (FM.java:81-86)
    const double *vu = a.V + (size_t)u * a.k;
    const double *vc = a.V + (size_t)((int64_t)a.n_users + a.n_items + (has_c ? c : 0)) * a.k;
    double *dst = A + (size_t)blockIdx.x * a.kp;
    double dot = 0.0;
    for (int f = threadIdx.x; f < a.kp; f += blockDim.x) {
        double v = 0.0;
        if (f < a.k) {
            v = vu[f];
            if (has_c) {
                v += a.xc * vc[f];
                dot += vu[f] * vc[f];
            }
        } else if (f == a.k) {
            v = 1.0;
        }
        dst[f] = v;
    }
    part[threadIdx.x] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int t = 0; t < (int)blockDim.x; ++t) s += part[t];
        double rc = *a.w0 + a.w[u];
        if (has_c) rc += a.xc * a.w[(int64_t)a.n_users + a.n_items + c] + a.xc * s;
        row_const[blockIdx.x] = rc;
    }
}

hipError_t rank_launch_fm_items(const RankFmArgs &a, const int32_t *cand, int nc, double *B, hipStream_t s) {
    if (nc <= 0) return hipSuccess;
    hipLaunchKernelGGL(rank_fm_items, dim3(nc), dim3(64), 0, s, a, cand, B);
    return hipGetLastError();
}
hipError_t rank_launch_fm_queries(const RankFmArgs &a, const int32_t *qu, const int32_t *qc, int nq, double *A, double *row_const,
                                  hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(rank_fm_queries, dim3(nq), dim3(64), 0, s, a, qu, qc, A, row_const);
    return hipGetLastError();
}

// ---- S[q][c] = <A[q,:], B[c,:]> + row_const[q]  (LDS-tiled, 64x64 tile, 4x4 per thread) -------------------------

template <typename T>
__global__ __launch_bounds__(256) void rank_gemm(const T *__restrict__ A, const T *__restrict__ B, const T *row_const,
                                                 T *__restrict__ S, int nq, int nc, int kp, const T *__restrict__ col_const) {
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
            if (c < nc) S[(size_t)q * nc + c] = (col_const ? acc[i][jj] + col_const[c] : acc[i][jj]) + rc;
        }
    }
}

// ---- fp32: the same contraction on the f32-input matrix cores ------------------------------------------------------
// v_mfma_f32_32x32x2_f32: exact f32 (a k-ordered fmaf chain), 64 FLOP/clk/SIMD = the f32 vector peak, reached with one
// VGPR per operand per lane.  Block = 128 queries x 128 candidates, BK = 16 (33 KB of LDS -> 3 blocks per CU; 32 measured 8% slower); 4 waves in a 2x2 grid, each owning a
// 64x64 patch = 2x2 MFMA tiles (64 accumulator VGPRs).  A/B tiles go through LDS k-major ([k][row]) so that the MFMA
// operand read (lane l: row l&31, k-slot l>>5) is one conflict-light ds_read_b32; the next tile's global loads are in
// flight while the current one is multiplied (register double-buffer, two LDS buffers, one barrier per BK step).
// Operands are padded by the builders: kp_pad % 32 == 0 (zero columns), row counts rounded up to 128.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef CMI_RG_BK
#define CMI_RG_BK 16
#endif
constexpr int RG_BM = 128, RG_BN = 128, RG_BK = CMI_RG_BK, RG_LDS = RG_BM + 2;
constexpr int RG_TPR = RG_BK / 4, RG_RPP = 256 / RG_TPR, RG_NP = RG_BM / RG_RPP; // staging: threads per row, rows per pass, passes // +2: spreads the transposing writes over banks

#ifndef CMI_RG_WAVES
#define CMI_RG_WAVES 4 // min waves per SIMD: 114 VGPRs instead of 140, four blocks per CU instead of three (23.70 -> 23.19 ms on the 270 K x 20 K case)
#endif
// tile_max (may be null): tile_max[q * nt64 + t] = the largest score the launch stored in row q among candidates [64 t, 64 t + 64)
// (NaN scores ignored).  A wave's 64 x 64 patch IS one such tile: the maximum costs a few DPP row operations per accumulator row.
// The selection uses it to skip tiles that cannot hold a candidate above a query's current N-th best (rank_topn_split_pruned).
// Four independent maxima over lanes 0-31 (left in lanes 16-31) and lanes 32-63 (in lanes 48-63) at once: v_max_f32 with the DPP
// operand on the instruction itself (row_ror 8 / 4 / 2 / 1: every lane of a 16-lane row holds the row's maximum; row_bcast:15 into
// rows 1 and 3 adds the previous row's).  Written as one asm block because the compiler's fmaxf costs three instructions per step
// (it quiets both operands first; the kernel runs in IEEE mode, where v_max_f32 already returns the non-NaN operand) -- 800
// instructions per tile instead of 200.  The four chains are interleaved, so a result is read by its next DPP step three
// instructions later (VALU write -> DPP read needs two wait states); s_nop 1 covers the values coming in.
__device__ __forceinline__ void rg_max32x4(float &a, float &b, float &c, float &d) {
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %2, %2, %2 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %3, %3, %3 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %2, %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %3, %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "v_max_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "v_max_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

__global__ __launch_bounds__(256, CMI_RG_WAVES) void rank_gemm_mfma_f32(const float *__restrict__ A, const float *__restrict__ B,
                                                          const float *__restrict__ row_const, float *__restrict__ S,
                                                          int nq, int nc, int kp_pad, int tiles_c, int n_tiles, const float *__restrict__ col_const,
                                                          float *__restrict__ tile_max, int nt64) {
    __shared__ float sA[2][RG_BK][RG_LDS];
    __shared__ float sB[2][RG_BK][RG_LDS];
    __shared__ float s_rc[RG_BM];
    // XCD-aware tile order: workgroup ids are dealt round-robin to the 8 XCDs, so give XCD x the x-th contiguous
    // eighth of the tile list (candidate tiles fastest): its L2 then sees one band of queries and sweeps the items.
    const int per = (n_tiles + 7) / 8;
    const int tile = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const int q0 = (tile / tiles_c) * RG_BM, c0 = (tile % tiles_c) * RG_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wq = (wave >> 1) * 64, wc = (wave & 1) * 64; // this wave's 64x64 patch inside the block tile
    const int lrow = tid / RG_TPR, lk = (tid % RG_TPR) * 4; // global->LDS staging: RG_TPR threads cover the BK k of one row
    const float *Ag = A + (size_t)(q0 + lrow) * kp_pad + lk;
    const float *Bg = B + (size_t)(c0 + lrow) * kp_pad + lk;
    f32x4 ra[RG_NP], rb[RG_NP];
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < RG_NP; ++i) {
            ra[i] = *reinterpret_cast<const f32x4 *>(Ag + (size_t)(RG_RPP * i) * kp_pad + k0);
            rb[i] = *reinterpret_cast<const f32x4 *>(Bg + (size_t)(RG_RPP * i) * kp_pad + k0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RG_NP; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sA[buf][lk + e][lrow + RG_RPP * i] = ra[i][e];
                sB[buf][lk + e][lrow + RG_RPP * i] = rb[i][e];
            }
    };
    gload(0);
    // the tile's 128 row constants go to LDS now (visible after the barrier below): the epilogue used to fetch each with a global load in
    // front of the row's stores, and on this part a wait for a load is also a wait for every older store -- 32 dependent round trips
    // per wavefront and tile, as long as the tile's whole matrix work (contraction 2.9 -> see profiles/r06_rank_epilogue.txt)
    if (tid < RG_BM) s_rc[tid] = q0 + tid < nq ? row_const[q0 + tid] : 0.f;
    lstore(0);
    __syncthreads();
    const int nk = kp_pad / RG_BK;
    const int mrow = lane & 31, mk = lane >> 5;
    for (int it = 0; it < nk; ++it) {
        const int buf = it & 1;
        if (it + 1 < nk) gload((it + 1) * RG_BK);
        // operand registers are double-buffered by hand: the ds_reads of k-step s+1 are issued before the four
        // MFMAs of step s (256 cycles of matrix-pipe work), so their latency never stalls the pipe
        float a0[2], a1[2], b0[2], b1[2];
        a0[0] = sA[buf][mk][wq + mrow];
        a1[0] = sA[buf][mk][wq + 32 + mrow];
        b0[0] = sB[buf][mk][wc + mrow];
        b1[0] = sB[buf][mk][wc + 32 + mrow];
#pragma unroll
        for (int st = 0; st < RG_BK / 2; ++st) {
            const int cur = st & 1, nxt = cur ^ 1;
            if (st + 1 < RG_BK / 2) {
                const int kk = 2 * (st + 1) + mk;
                a0[nxt] = sA[buf][kk][wq + mrow];
                a1[nxt] = sA[buf][kk][wq + 32 + mrow];
                b0[nxt] = sB[buf][kk][wc + mrow];
                b1[nxt] = sB[buf][kk][wc + 32 + mrow];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[cur], b0[cur], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[cur], b1[cur], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cur], b0[cur], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cur], b1[cur], acc[1][1], 0, 0, 0);
            // pin the interleave (the machine scheduler otherwise sinks the reads behind the MFMAs to save 4 VGPRs):
            // [DS reads of the next step] then [4 MFMA]
            if (st == 0) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            else if (st + 1 < RG_BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (st == RG_BK / 4 && it + 1 < nk) { // mid-tile: the prefetched next tile goes to the idle LDS buffer
                lstore(buf ^ 1);
                __builtin_amdgcn_sched_group_barrier(0x200, 4 * RG_NP, 0);
            }
        }
        if (it + 1 < nk) __syncthreads(); // the other buffer was last read one barrier ago
    }
    // D layout of the 32x32 MFMA: column (B index) = lane & 31, row (A index) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float cc[2] = {0.f, 0.f};
    if (col_const) {
#pragma unroll
        for (int j = 0; j < 2; ++j) cc[j] = col_const[min(c0 + wc + 32 * j + mrow, nc - 1)];
    }
    float rowmax = -INFINITY; // lane l: the maximum of patch row l over the wave's 64 candidates (collected below, stored once)
    // FULL (block-uniform): the whole 128 x 128 tile lies inside the slab -- no per-element bounds, the 64 stores of a wavefront go out
    // back to back; the edge tiles take the same code with the bounds
    auto epilogue = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r4 = 0; r4 < 16; r4 += 4) {
                float m[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int r = r4 + d;
                    const int ql = wq + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * mk, q = q0 + ql;
                    const float rc = s_rc[ql]; // (0 for rows past the end)
                    m[d] = -INFINITY; // (no `continue` for rows past the end: the cross-lane maximum below needs every lane)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int c = c0 + wc + 32 * j + mrow;
                        const float val = (col_const ? acc[i][j][r] + cc[j] : acc[i][j][r]) + rc;
                        if (FULL || (q < nq && c < nc)) {
                            S[(size_t)q * nc + c] = val;
                            m[d] = val > m[d] ? val : m[d]; // (a NaN score is never taken: it cannot enter a list either)
                        }
                    }
                }
                if (tile_max) { // wave-uniform
                    rg_max32x4(m[0], m[1], m[2], m[3]);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int r = r4 + d, row = 32 * i + (r & 3) + 8 * (r >> 2); // patch row of the lanes with mk = 0; mk = 1: + 4
                        const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m[d]), 31));
                        const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m[d]), 63));
                        rowmax = lane == row ? lo : (lane == row + 4 ? hi : rowmax);
                    }
                }
            }
    };
    if (q0 + RG_BM <= nq && c0 + RG_BN <= nc) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    if (tile_max && q0 + wq + lane < nq && c0 + wc < nc) tile_max[(size_t)(q0 + wq + lane) * nt64 + ((c0 + wc) >> 6)] = rowmax;
}

// tile maxima of a finished score slab (only where the MFMA contraction did not produce them: the VALU fallback): a wave per (row, tile)
template <typename T>
__global__ __launch_bounds__(256) void rank_tile_max(const T *__restrict__ S, int nq, int nc, int nt64, T *__restrict__ tile_max) {
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (int64_t)nq * nt64) return;
    const int q = (int)(w / nt64), t = (int)(w % nt64), c = t * 64 + (int)(threadIdx.x & 63);
    T m = c < nc ? S[(size_t)q * nc + c] : (T)-INFINITY;
    if (m != m) m = (T)-INFINITY; // NaN scores never enter a list
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const T x = __shfl_xor(m, o, 64);
        m = x > m ? x : m;
    }
    if ((threadIdx.x & 63) == 0) tile_max[w] = m;
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

// ---- top-N, single pass (topn <= 64): the list lives in the wave's registers, lane r = rank r ------------------------
// The row is streamed once, 64 x RT_U scores per step.  A score enters only if it beats the current N-th best (or the
// list is not full), so after the first few steps almost every step is a pure load + compare + ballot; expected
// insertions per row ~ N ln(n/N).  Scores are consumed in ascending candidate order and an equal score never
// overtakes an earlier one -> the reference's stable descending sort over its candidate order.
constexpr int RT_U = 8;

template <typename T>
__device__ __forceinline__ T lane_bcast(T v, int l) { return __shfl(v, l, 64); }

template <typename T>
__global__ __launch_bounds__(256) void rank_topn_stream(const T *__restrict__ S, int nq, int nc, double thold, int topn,
                                                        int32_t *out_idx, double *out_score, int32_t *out_count,
                                                        int q_base) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const T *row = S + (size_t)q * nc;
    T lv = -INFINITY; // rank `lane` of the list
    int li = -1;
    int count = 0;
    T t = -INFINITY;  // the N-th best once the list is full
    for (int base = 0; base < nc; base += 64 * RT_U) {
        T v[RT_U];
#pragma unroll
        for (int u = 0; u < RT_U; ++u) {
            const int c = base + u * 64 + lane;
            v[u] = c < nc ? row[c] : (T)-INFINITY;
        }
#pragma unroll
        for (int u = 0; u < RT_U; ++u) {
            // `score > threshold`, not NaN (Recommender.java:808-812); masked items are -inf
            unsigned long long m = __ballot((double)v[u] > thold && v[u] > -INFINITY && (count < topn || v[u] > t));
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const T cv = lane_bcast(v[u], l);
                if (count == topn && !(cv > t)) continue;
                const int pos = __popcll(__ballot(lane < count && lv >= cv));
                const T uv = __shfl_up(lv, 1, 64);
                const int ui = __shfl_up(li, 1, 64);
                if (lane > pos) {
                    lv = uv;
                    li = ui;
                }
                if (lane == pos) {
                    lv = cv;
                    li = base + u * 64 + l;
                }
                if (count < topn) ++count;
                if (count == topn) t = lane_bcast(lv, topn - 1);
            }
        }
    }
    if (lane < count) {
        out_idx[(size_t)(q_base + q) * topn + lane] = li;
        out_score[(size_t)(q_base + q) * topn + lane] = (double)lv;
    }
    if (lane == 0) out_count[q_base + q] = count;
}


// ---- the split form of the MF family's scores (round 4; fp32 state) ----------------------------------------------------------------------
// For BiasedMF / PMF / CAMF_C / CAMF_CI / CAMF_CU / CAMF_CUCI the item-dependent part of predict(u, j, c) is a SUM of a part that
// depends on (user, item) only and a part that depends on (context, item) only:
//     score[q][j] = ( <P[u_q], Q[j]> + itemBias[j] )  +  sum_{cond in c_q} icBias[j][cond]  +  row_const[q]
//                 =        S1[u_q][j]               +          S2[c_q][j]                  +  rc[q]
// A test user appears in several contexts (5.4 queries per user in the bench's case) and far fewer distinct contexts exist than queries,
// so contracting per QUERY repeats the k-long user part once per context.  The split form contracts S1 once per distinct query USER
// (operand length k + 1) and S2 once per distinct CONTEXT (operand length n_conds), and the selection adds the two rows while it streams
// them: 7x fewer matrix-core flops and a 5x smaller slab in the bench's case.  fp32 only (a different association of the same sum than
// the per-query dot product: inside the fp32 tolerance of the ranking tests; the fp64 verification path keeps the per-query contraction).
template <typename T>
__global__ void rank_build_ic_items(const T *__restrict__ icBias, const int32_t *__restrict__ cand, T *__restrict__ B2, int n_conds, int kp2) {
    const int c = blockIdx.x, j = cand[c];
    for (int f = threadIdx.x; f < kp2; f += blockDim.x) B2[(size_t)c * kp2 + f] = f < n_conds ? icBias[(size_t)j * n_conds + f] : (T)0;
}
// one-hot rows of the distinct contexts' condition lists
template <typename T>
__global__ void rank_build_ctx_rows(const int32_t *__restrict__ ctx_ptr, const int32_t *__restrict__ ctx_conds, const int32_t *__restrict__ dctx,
                                    T *__restrict__ A2, int kp2) {
    const int r = blockIdx.x, c = dctx[r];
    for (int f = threadIdx.x; f < kp2; f += blockDim.x) A2[(size_t)r * kp2 + f] = (T)0;
    __syncthreads();
    for (int p = ctx_ptr[c] + threadIdx.x; p < ctx_ptr[c + 1]; p += blockDim.x) A2[(size_t)r * kp2 + ctx_conds[p]] = (T)1;
}
// rc[q]: every predict() term that does not depend on the item (as rank_build_queries computes it)
template <typename T>
__global__ void rank_query_consts(RankQueryArgs<T> a, int nq) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int u = a.qu[q], c = a.qc[q];
    T rc = (T)a.gm;
    if (a.userBias) rc += a.userBias[u];
    if (a.ctx_ptr)
        for (int p = a.ctx_ptr[c]; p < a.ctx_ptr[c + 1]; ++p) {
            const int cond = a.ctx_conds[p];
            if (a.ucBias) rc += a.ucBias[(size_t)u * a.n_conds + cond];
            if (a.condBias) rc += a.condBias[cond];
        }
    a.row_const[q] = rc;
}

// top-N of query q over  S1[group of q] + S2[context of q] + rc[q]  with the query's exclusion list (ascending candidate positions)
// skipped on the fly -- the slab rows are shared between queries, so nothing is masked in place.  Same list discipline as
// rank_topn_stream (lane r = rank r, an equal score never overtakes an earlier one).
template <typename T>
__global__ __launch_bounds__(256) void rank_topn_split(const T *__restrict__ S1, const T *__restrict__ S2, const T *__restrict__ rc,
                                                       const int32_t *__restrict__ q_group, const int32_t *__restrict__ q_dctx, int g_base, int q0,
                                                       int nq, int nc, const int64_t *__restrict__ excl_ptr, const int32_t *__restrict__ excl_idx,
                                                       T tf, int topn, int32_t *out_idx, double *out_score, int32_t *out_count) {
    // tf = the largest T <= the rating threshold: (double)x > threshold  <=>  x > tf, for every x of type T
    const int lane = threadIdx.x & 63;
    // XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs, so XCD x takes the x-th contiguous eighth of the queries -- a
    // user's queries (consecutive) then share one XCD's L2 for their S1 row instead of fetching it into two (A/B on one box, two runs
    // each: 4.22 / 4.21 ms per evaluation with workgroup b on queries 4b .., 3.50 / 3.51 ms with this order)
    const int per = (int)gridDim.x / 8; // (the grid is a multiple of 8 workgroups)
    const int blk = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    const int q = q0 + blk * 4 + (threadIdx.x >> 6);
    if (q >= q0 + nq) return;
    const T *row1 = S1 + (size_t)(q_group[q] - g_base) * nc;
    const T *row2 = S2 ? S2 + (size_t)q_dctx[q] * nc : nullptr;
    const T c0 = rc[q];
    int64_t ep = excl_ptr[q];
    const int64_t ee = excl_ptr[q + 1];
    int next_excl = ep < ee ? excl_idx[ep] : 0x7fffffff;
    T lv = -INFINITY;
    int li = -1;
    int count = 0;
    T t = -INFINITY;
    // ONE comparison per score: `x > thr` with thr = tf while the list fills and the N-th best once it is full (every list entry is
    // > tf, so max(tf, t) = t) -- false for NaN, for masked / out-of-range entries (-inf) and for `score > threshold` failing
    // (Recommender.java:808-812)
    T thr = tf;
    T n1[RT_U], n2[RT_U];
    auto request = [&](int base) {
        if (base + 64 * RT_U <= nc) { // a whole step
#pragma unroll
            for (int u = 0; u < RT_U; ++u) n1[u] = row1[base + u * 64 + lane];
            if (row2) {
#pragma unroll
                for (int u = 0; u < RT_U; ++u) n2[u] = row2[base + u * 64 + lane];
            }
        } else {
#pragma unroll
            for (int u = 0; u < RT_U; ++u) {
                const int c = base + u * 64 + lane;
                n1[u] = c < nc ? row1[c] : (T)-INFINITY;
                n2[u] = c < nc && row2 ? row2[c] : (T)0;
            }
        }
    };
    for (int base = 0; base < nc; base += 64 * RT_U) {
        T v[RT_U];
        request(base);
        if (row2) {
#pragma unroll
            for (int u = 0; u < RT_U; ++u) v[u] = (n1[u] + n2[u]) + c0;
        } else {
#pragma unroll
            for (int u = 0; u < RT_U; ++u) v[u] = n1[u] + c0;
        }
        // (requesting the next step's rows here, behind this step's arithmetic, was measured: 4.4 against 4.3 ms -- no gain, the kernel
        // waits on the memory system's throughput, not on a single request's latency; profiles/r04_rank_selection_forms.txt)
        // the (few) already-rated items of this query that fall into this step's 64 * RT_U candidates: wave-uniform walk
        while (next_excl < base + 64 * RT_U) {
            const int off = next_excl - base;
#pragma unroll
            for (int u = 0; u < RT_U; ++u)
                if (off >= u * 64 && off < (u + 1) * 64 && lane == off - u * 64) v[u] = -INFINITY;
            ++ep;
            next_excl = ep < ee ? excl_idx[ep] : 0x7fffffff;
        }
#pragma unroll
        for (int u = 0; u < RT_U; ++u) {
            unsigned long long m = __ballot(v[u] > thr);
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const T cv = lane_bcast(v[u], l);
                if (count == topn && !(cv > t)) continue;
                const int pos = __popcll(__ballot(lane < count && lv >= cv));
                const T uv = __shfl_up(lv, 1, 64);
                const int ui = __shfl_up(li, 1, 64);
                if (lane > pos) {
                    lv = uv;
                    li = ui;
                }
                if (lane == pos) {
                    lv = cv;
                    li = base + u * 64 + l;
                }
                if (count < topn) ++count;
                if (count == topn) thr = t = lane_bcast(lv, topn - 1);
            }
        }
    }
    if (lane < count) {
        out_idx[(size_t)q * topn + lane] = li;
        out_score[(size_t)q * topn + lane] = (double)lv;
    }
    if (lane == 0) out_count[q] = count;
}

// The same selection with TILE PRUNING (round 6).  M1[g][t] / M2[c][t] = the largest S1 / S2 value of the row over candidates
// [64 t, 64 t + 64) (written by the contractions' epilogues).  ub = (M1 + M2) + c0 is an upper bound of every score of the tile in the
// SAME floating-point operations as the score itself -- n1 <= M1 and n2 <= M2 give fl(n1 + n2) <= fl(M1 + M2), and adding c0 keeps
// the order (rounding is monotone) -- so a tile with !(ub > thr) cannot contain a candidate that passes `v > thr` and is neither
// loaded nor scanned.  thr only rises while a row is walked, tiles are still visited in ascending order, lanes in ascending order
// inside a tile: the lists are those of rank_topn_split, entry for entry (tests/test_gpu_ranking.py compares the two forms bit for bit).
// A top-10 list's threshold sits ~3.3 sigma out, a 64-candidate tile's maximum ~2.4: on the bench's data 60 % of the tiles are skipped.
// Lane l of a chunk holds the bound of tile tb + l; the tiles that pass are fetched PU at a time (2 PU loads in flight, as before).
// v of lane l for a WAVE-UNIFORM l (v_readlane with the lane number in an SGPR; __shfl goes through LDS)
__device__ __forceinline__ float lane_pick(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double lane_pick(double v, int l) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <typename T>
__global__ __launch_bounds__(256) void rank_topn_split_pruned(const T *__restrict__ S1, const T *__restrict__ S2, const T *__restrict__ M1,
                                                              const T *__restrict__ M2, int nt64, const T *__restrict__ rc,
                                                              const int32_t *__restrict__ q_group, const int32_t *__restrict__ q_dctx, int g_base, int q0,
                                                              int nq, int nc, const int64_t *__restrict__ excl_ptr, const int32_t *__restrict__ excl_idx,
                                                              T tf, int topn, int32_t *out_idx, double *out_score, int32_t *out_count) {
#ifndef CMI_RT_PU
#define CMI_RT_PU 8
#endif
    constexpr int PU = CMI_RT_PU;
    const int lane = threadIdx.x & 63;
    const int per = (int)gridDim.x / 8; // XCD-aware order, as in rank_topn_split
    const int blk = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    // the wave's query index as a SCALAR (readfirstlane of the wave number): the row pointers, the exclusion cursor and its reads then live
    // in SGPRs / scalar loads -- as VGPR values the compiler walked the exclusion list with vector loads and `s_waitcnt vmcnt(0)` in front
    // of every tile (which also drains the round's tile loads), and broadcast the tile bound through LDS (ds_bpermute)
    const int q = q0 + blk * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (q >= q0 + nq) return;
    const size_t g = (size_t)(q_group[q] - g_base);
    const T *row1 = S1 + g * nc;
    const T *row2 = S2 ? S2 + (size_t)q_dctx[q] * nc : nullptr;
    const T *m1 = M1 + g * nt64;
    const T *m2 = S2 ? M2 + (size_t)q_dctx[q] * nt64 : nullptr;
    const T c0 = rc[q];
    int64_t ep = excl_ptr[q];
    const int64_t ee = excl_ptr[q + 1];
    int next_excl = ep < ee ? excl_idx[ep] : 0x7fffffff;
    T lv = -INFINITY;
    int li = -1;
    int count = 0;
    T t = -INFINITY;
    T thr = tf;
    // Lane l holds the bound of tile tb + l of the current chunk of 64 tiles (the next chunk's bounds are requested a chunk ahead).  The
    // chunk's tiles that pass are taken in ROUNDS of up to PU: all of a round's 2 PU loads are issued back to back before its first tile is
    // scanned -- as many in flight per wave as the plain selection has.  Keep the round's fill loop free of other loads and loops: with
    // the chunk advance inside it the compiler waited after every load (4.0 instead of 2.6 ms); two rounds in flight (the next round's
    // loads issued before this one is scanned) cost 82 VGPRs and gained nothing (3.8 ms).
    auto bounds = [&](int base) -> T {
        T u = -INFINITY;
        if (base + lane < nt64) u = m2 ? (m1[base + lane] + m2[base + lane]) + c0 : m1[base + lane] + c0;
        return u;
    };
    T ubn = bounds(0);
    for (int tb = 0; tb < nt64; tb += 64) {
        const T ub = ubn;
        ubn = bounds(tb + 64);
        unsigned long long mask = __ballot(ub > thr);
        while (mask) {
            int tile[PU];
            T n1[PU], n2[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) { // the next PU tiles that pass (wave-uniform), all their loads issued before the first is used
                tile[u] = -1;
                if (mask) {
                    tile[u] = tb + __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    // (no per-lane bounds around the loads -- as exec-masked branches they were half of the fill's instructions: the row's last
                    // tile may read up to 63 elements past the row, which the slabs' allocations cover (RANK_SLAB_SLACK); the scan masks them)
                    const int c = tile[u] * 64 + lane;
                    n1[u] = row1[c];
                    n2[u] = row2 ? row2[c] : (T)0;
                }
            }
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                if (tile[u] < 0) continue;
                if (!(lane_pick(ub, tile[u] - tb) > thr)) continue; // the N-th best has risen past this tile's bound meanwhile (uniform)
                const int base = tile[u] * 64;
                T v = row2 ? (n1[u] + n2[u]) + c0 : n1[u] + c0;
                if (base + 64 > nc) v = base + lane < nc ? v : (T)-INFINITY; // (wave-uniform test) the row's last, partial tile
                while (next_excl < base) { // already-rated items inside skipped tiles
                    ++ep;
                    next_excl = ep < ee ? excl_idx[ep] : 0x7fffffff;
                }
                while (next_excl < base + 64) {
                    if (lane == next_excl - base) v = -INFINITY;
                    ++ep;
                    next_excl = ep < ee ? excl_idx[ep] : 0x7fffffff;
                }
                unsigned long long m = __ballot(v > thr);
                while (m) {
                    const int l = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const T cv = lane_pick(v, l);
                    if (count == topn && !(cv > t)) continue;
                    const int pos = __popcll(__ballot(lane < count && lv >= cv));
                    const T uv = __shfl_up(lv, 1, 64);
                    const int ui = __shfl_up(li, 1, 64);
                    if (lane > pos) {
                        lv = uv;
                        li = ui;
                    }
                    if (lane == pos) {
                        lv = cv;
                        li = base + l;
                    }
                    if (count < topn) ++count;
                    if (count == topn) thr = t = lane_pick(lv, topn - 1);
                }
            }
            mask &= __ballot(ub > thr); // drop the chunk's remaining tiles that no longer pass
        }
    }
    if (lane < count) {
        out_idx[(size_t)q * topn + lane] = li;
        out_score[(size_t)q * topn + lane] = (double)lv;
    }
    if (lane == 0) out_count[q] = count;
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
hipError_t rank_launch_gemm(const T *A, const T *B, const T *row_const, T *S, int nq, int nc, int kp, hipStream_t s, const T *col_const, T *tile_max) {
    if (nq <= 0 || nc <= 0) return hipSuccess;
    const int nt64 = (nc + 63) / 64;
    static const bool force_valu = cmi_exp_env("CMI_RANK_VALU") != nullptr; // A/B experiments only
    if constexpr (sizeof(T) == 4) {
        if (!force_valu && kp % RG_BK == 0) {
            const int tiles_c = (nc + RG_BN - 1) / RG_BN, n_tiles = tiles_c * ((nq + RG_BM - 1) / RG_BM);
            hipLaunchKernelGGL(rank_gemm_mfma_f32, dim3(((n_tiles + 7) / 8) * 8), dim3(256), 0, s, (const float *)A,
                               (const float *)B, (const float *)row_const, (float *)S, nq, nc, kp, tiles_c, n_tiles, (const float *)col_const,
                               (float *)tile_max, nt64);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL(rank_gemm<T>, dim3((nc + 63) / 64, (nq + 63) / 64), dim3(256), 0, s, A, B, row_const, S, nq, nc, kp, col_const);
    if (tile_max) hipLaunchKernelGGL(rank_tile_max<T>, dim3((unsigned)(((int64_t)nq * nt64 + 3) / 4)), dim3(256), 0, s, (const T *)S, nq, nc, nt64, tile_max);
    return hipGetLastError();
}
template <typename T>
hipError_t rank_launch_score(const T *A, const T *B, const T *row_const, T *S, int nq, int nc, int kp,
                             const int64_t *excl_ptr, const int32_t *excl_idx, int q_base, double thold, int topn,
                             int32_t *out_idx, double *out_score, int32_t *out_count, hipStream_t s) {
    if (nq <= 0 || nc <= 0) return hipSuccess;
    if (hipError_t e = rank_launch_gemm<T>(A, B, row_const, S, nq, nc, kp, s, nullptr, nullptr)) return e;
    hipLaunchKernelGGL(rank_mask<T>, dim3(nq), dim3(64), 0, s, S, nc, excl_ptr, excl_idx, q_base, nq);
    if (topn <= 64)
        hipLaunchKernelGGL(rank_topn_stream<T>, dim3((nq + 3) / 4), dim3(256), 0, s, (const T *)S, nq, nc, thold, topn, out_idx,
                           out_score, out_count, q_base);
    else // long lists: one extraction pass per rank
        hipLaunchKernelGGL(rank_topn<T>, dim3((nq + 3) / 4), dim3(256), 0, s, S, nq, nc, thold, topn, out_idx, out_score,
                           out_count, q_base);
    return hipGetLastError();
}


hipError_t rank_launch_split_operands(const RankSplitArgs &a, hipStream_t s) {
    // B1 = Q[j] for the candidates (k columns, zero-padded to kp1); itemBias[j] goes to colc: the contraction adds it at the end
    RankItemsArgs<float> ia{a.Q, a.itemBias, nullptr, a.cand, a.B1, a.nc, a.k, a.kp1, 0, a.colc};
    if (hipError_t e = rank_launch_build_items<float>(ia, s)) return e;
    if (a.icBias) {
        hipLaunchKernelGGL(rank_build_ic_items<float>, dim3(a.nc), dim3(64), 0, s, a.icBias, a.cand, a.B2, a.n_conds, a.kp2);
        hipLaunchKernelGGL(rank_build_ctx_rows<float>, dim3(a.n_dctx), dim3(64), 0, s, a.ctx_ptr, a.ctx_conds, a.dctx, a.A2, a.kp2);
    }
    RankQueryArgs<float> qa{nullptr, a.userBias, a.ucBias, a.condBias, a.qu, a.qc, a.ctx_ptr, a.ctx_conds, nullptr, a.rc, a.gm, a.k, a.kp1, a.n_conds, 0};
    hipLaunchKernelGGL(rank_query_consts<float>, dim3((a.nq + 255) / 256), dim3(256), 0, s, qa, a.nq);
    return hipGetLastError();
}
// A1 rows = [P[u] | 1] of the distinct query users [g0, g0 + n) (scratch_rc receives the builder's per-row constant, unused here)
hipError_t rank_launch_split_users(const RankSplitArgs &a, const int32_t *d_group_user, int n, float *A1, float *scratch_rc, hipStream_t s) {
    RankQueryArgs<float> qa{a.P, nullptr, nullptr, nullptr, d_group_user, d_group_user, nullptr, nullptr, A1, scratch_rc, 0.0, a.k, a.kp1, a.n_conds, 0};
    return rank_launch_build_queries<float>(qa, n, s);
}
hipError_t rank_launch_split_select(const float *S1, const float *S2, const RankSplitArgs &a, const int32_t *q_group, const int32_t *q_dctx, int g_base,
                                    int q0, int nq, const int64_t *excl_ptr, const int32_t *excl_idx, double thold, int topn, int32_t *out_idx,
                                    double *out_score, int32_t *out_count, hipStream_t s, const float *M1, const float *M2) {
    if (nq <= 0) return hipSuccess;
    float tf = (float)thold; // round to nearest, then down to the largest float <= thold (NaN stays NaN: nothing passes, as before)
    if ((double)tf > thold) tf = nextafterf(tf, -INFINITY);
    const int nblk = (nq + 3) / 4;
    if (M1 && (M2 || !S2)) { // the tile maxima are there: the pruning form
        hipLaunchKernelGGL(rank_topn_split_pruned<float>, dim3((nblk + 7) / 8 * 8), dim3(256), 0, s, S1, S2, M1, M2, (a.nc + 63) / 64, (const float *)a.rc, q_group,
                           q_dctx, g_base, q0, nq, a.nc, excl_ptr, excl_idx, tf, topn, out_idx, out_score, out_count);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(rank_topn_split<float>, dim3((nblk + 7) / 8 * 8), dim3(256), 0, s, S1, S2, (const float *)a.rc, q_group, q_dctx, g_base, q0, nq, a.nc,
                       excl_ptr, excl_idx, tf, topn, out_idx, out_score, out_count);
    return hipGetLastError();
}

#define CMI_INST(T)                                                                                                    \
    template hipError_t rank_launch_build_items<T>(const RankItemsArgs<T> &, hipStream_t);                             \
    template hipError_t rank_launch_build_queries<T>(const RankQueryArgs<T> &, int, hipStream_t);                      \
    template hipError_t rank_launch_gemm<T>(const T *, const T *, const T *, T *, int, int, int, hipStream_t, const T *, T *);   \
    template hipError_t rank_launch_score<T>(const T *, const T *, const T *, T *, int, int, int, const int64_t *,     \
                                             const int32_t *, int, double, int, int32_t *, double *, int32_t *, hipStream_t);
CMI_INST(float)
CMI_INST(double)
#undef CMI_INST

} // namespace cmi

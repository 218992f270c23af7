// fm_kernels.hip -- gfx950 kernels for the reference's FM recommender
// (src/carskit/alg/cars/adaptation/dependent/FM.java:115-220): an ALS / coordinate-descent sweep, NOT SGD.
//
// The reference walks all p = numUsers+numItems+numConditions coordinates one after another and, for each,
// loops over ALL ratings with a dense feature vector (O(size*p*k) per sweep).  Every rating has at most three
// non-zero features -- its user (value 1), its item (1) and, if its context-combination id c is < numConditions
// (the reference's index quirk, FM.java:81-86), feature numUsers+numItems+c with value 1/numContextDims.
// Rows whose feature l is zero contribute exactly 0 to the numerator and exactly `reg` to the denominator of
// coordinate l and are not touched by its error update.  So
//   * a coordinate only needs the ratings in its support (CSR lists per field, built once on the host);
//   * the coordinates of one FIELD (all users / all items / all context features) have pairwise disjoint
//     supports, so their sequential updates commute exactly and run in parallel;
//   * the denominator is sum_{support} h^2 + size*reg.
// One sweep = 1 (w0) + 3 (w: users, items, contexts) + 3k (V, per factor) phases.  Each phase is a segmented
// reduction (num, den per coordinate) followed by the coordinate update and the error / Q update on the support.
// fp64 throughout (the reference's precision); sums are tree-reduced, so results match the sequential Java
// sums to rounding (tests hold 1e-9).  HBM-bound gather/scatter over errors[] and one Q column: no MFMA.
//
// Split reduce/apply kernels exist so a multi-GPU host can all-reduce (num, den) between them; the fused
// kernel is used on a single GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fm_kernels.hpp"

namespace cmi {

template <int BLOCK>
__device__ __forceinline__ double block_sum(double x, double *lds) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    if (BLOCK == 64) return x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads(); // lds reuse
    if (lane == 0) lds[wave] = x;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) s += lds[w];
    return s;
}

// feature value of field: users/items 1, context 1/numContextDims
__device__ __forceinline__ double field_x(const FmArgs &a, int field) { return field == 2 ? a.xc : 1.0; }

// MODE 0: reduce only (writes part[l], part[count+l]); MODE 1: apply only (reads part); MODE 2: fused.
// f < 0: linear weights w; f >= 0: factor column f of V.
template <int BLOCK, int MODE>
__global__ __launch_bounds__(BLOCK) void fm_field_kernel(FmArgs a, int field, int f) {
    __shared__ double lds[BLOCK / 64 + 2];
    const int l = blockIdx.x; // coordinate within the field
    const int64_t b = a.sup_off[field][l], e = a.sup_off[field][l + 1];
    const int32_t *sup = a.sup[field];
    const int64_t base = field == 0 ? 0 : (field == 1 ? a.n_users : (int64_t)a.n_users + a.n_items);
    double *theta_p = f < 0 ? a.w + base + l : a.V + (size_t)(base + l) * a.k + f;
    const double theta = *theta_p;
    const double x = field_x(a, field);
    double *q = f < 0 ? nullptr : a.Qt + (size_t)f * a.n;
    double num = 0.0, den = 0.0;
    if (MODE != 1) {
        for (int64_t s = b + threadIdx.x; s < e; s += BLOCK) {
            const int32_t i = sup ? sup[s] : (int32_t)s; // users: the storage order IS sorted by user
            const double h = f < 0 ? x : x * q[i] - x * x * theta;
            num += (a.err[i] - theta * h) * h;
            den += h * h;
        }
        num = block_sum<BLOCK>(num, lds);
        den = block_sum<BLOCK>(den, lds);
        if (MODE == 0) {
            if (threadIdx.x == 0) {
                a.part[l] = num;
                a.part[a.field_count[field] + l] = den;
            }
            return;
        }
    } else {
        num = a.part[l];
        den = a.part[a.field_count[field] + l];
    }
    const double reg = f < 0 ? a.regLw : a.regLf;
    const double upd = 0.0 - num / (den + (double)a.global_size * reg);
    const double delta = upd - theta;
    for (int64_t s = b + threadIdx.x; s < e; s += BLOCK) {
        const int32_t i = sup ? sup[s] : (int32_t)s;
        a.err[i] = a.err[i] + delta * x;
        if (q) q[i] = q[i] + delta * x;
    }
    if (BLOCK > 64) __syncthreads();
    if (threadIdx.x == 0) *theta_p = upd;
}

// w0 phase, reduce: part[0] = sum(err_i - w0) over the local ratings (fixed two-stage tree)
__global__ __launch_bounds__(256) void fm_w0_reduce1(FmArgs a, double *scratch) {
    __shared__ double lds[6];
    const int64_t chunk = (a.n + gridDim.x - 1) / gridDim.x;
    const int64_t b = (int64_t)blockIdx.x * chunk, e = (b + chunk) < a.n ? (b + chunk) : a.n;
    const double w0 = *a.w0;
    double s = 0.0;
    for (int64_t i = b + threadIdx.x; i < e; i += 256) s += a.err[i] - w0;
    s = block_sum<256>(s, lds);
    if (threadIdx.x == 0) scratch[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void fm_w0_reduce2(FmArgs a, const double *scratch, int nblk) {
    __shared__ double lds[6];
    double s = (int)threadIdx.x < nblk ? scratch[threadIdx.x] : 0.0;
    s = block_sum<256>(s, lds);
    if (threadIdx.x == 0) {
        a.part[0] = s;
        a.part[1] = 0.0;
    }
}
// w0 phase, apply: w0' = -part[0]/(size + regLw); err_i += w0' - w0   (FM.java:153-169)
__global__ __launch_bounds__(256) void fm_w0_apply(FmArgs a) {
    const double w0 = *a.w0;
    const double upd = 0.0 - a.part[0] / ((double)a.global_size + a.regLw);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256)
        a.err[i] = a.err[i] + upd - w0;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.part[2] = upd; // committed to *w0 by fm_w0_commit after all blocks read w0
}
__global__ void fm_w0_commit(FmArgs a) { *a.w0 = a.part[2]; }

// pre-pass (FM.java:117-146): errors[i] = r_i - predict(i), Q[i][f] = sum_l V[l][f] x_il.  One wave per rating.
__global__ __launch_bounds__(256) void fm_init_kernel(FmArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < a.n; i += stride) {
        const int u = a.u[i], j = a.j[i], c = a.ctx[i];
        const bool has_c = c >= 0 && c < a.n_conds;
        const double *vu = a.V + (size_t)u * a.k, *vj = a.V + (size_t)(a.n_users + j) * a.k;
        const double *vc = a.V + (size_t)(a.n_users + a.n_items + (has_c ? c : 0)) * a.k;
        double pair = 0.0;
        for (int f = lane; f < a.k; f += 64) {
            const double d0 = vu[f], d1 = vj[f], d2 = has_c ? vc[f] * a.xc : 0.0;
            double s1 = 0.0 + d0;
            s1 += d1;
            double s2 = d0 * d0 + d1 * d1;
            if (has_c) {
                s1 += d2;
                s2 += d2 * d2;
            }
            a.Qt[(size_t)f * a.n + i] = s1;
            pair += s1 * s1 - s2;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) pair += __shfl_xor(pair, m, 64);
        if (lane == 0) {
            double pred = *a.w0 + a.w[u];
            pred += a.w[a.n_users + j];
            if (has_c) pred += a.w[a.n_users + a.n_items + c] * a.xc;
            a.err[i] = a.r[i] - (pred + 0.5 * pair);
        }
    }
}

// FM.predict (FM.java:93-113) for arbitrary tuples; one wave per tuple
__global__ __launch_bounds__(256) void fm_predict_kernel(FmArgs a, int64_t n, const int32_t *tu, const int32_t *tj,
                                                         const int32_t *tc, int bound, double lo, double hi,
                                                         double *out) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += stride) {
        const int u = tu[i], j = tj[i], c = tc[i];
        const bool has_c = c >= 0 && c < a.n_conds;
        const double *vu = a.V + (size_t)u * a.k, *vj = a.V + (size_t)(a.n_users + j) * a.k;
        const double *vc = a.V + (size_t)(a.n_users + a.n_items + (has_c ? c : 0)) * a.k;
        double pair = 0.0;
        for (int f = lane; f < a.k; f += 64) {
            const double d0 = vu[f], d1 = vj[f], d2 = has_c ? vc[f] * a.xc : 0.0;
            const double s1 = (d0 + d1) + d2;
            pair += s1 * s1 - ((d0 * d0 + d1 * d1) + d2 * d2);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) pair += __shfl_xor(pair, m, 64);
        if (lane == 0) {
            double pred = *a.w0 + a.w[u];
            pred += a.w[a.n_users + j];
            if (has_c) pred += a.w[a.n_users + a.n_items + c] * a.xc;
            pred += 0.5 * pair;
            if (bound) {
                if (pred > hi) pred = hi;
                if (pred < lo) pred = lo;
            }
            out[i] = pred;
        }
    }
}

// ---- launchers ------------------------------------------------------------------------------------------

template <int MODE>
static hipError_t launch_field_mode(const FmArgs &a, int field, int f, hipStream_t s) {
    const int count = a.field_count[field];
    if (count <= 0) return hipSuccess;
    if (field == 2) // few coordinates with very long supports: a whole 1024-thread workgroup each
        hipLaunchKernelGGL((fm_field_kernel<1024, MODE>), dim3(count), dim3(1024), 0, s, a, field, f);
    else
        hipLaunchKernelGGL((fm_field_kernel<64, MODE>), dim3(count), dim3(64), 0, s, a, field, f);
    return hipGetLastError();
}

hipError_t fm_launch_field(const FmArgs &a, int field, int f, int mode, hipStream_t s) {
    switch (mode) {
    case 0: return launch_field_mode<0>(a, field, f, s);
    case 1: return launch_field_mode<1>(a, field, f, s);
    default: return launch_field_mode<2>(a, field, f, s);
    }
}

hipError_t fm_launch_w0_reduce(const FmArgs &a, double *scratch, hipStream_t s) {
    int nblk = (int)((a.n + 65535) / 65536);
    if (nblk < 1) nblk = 1;
    if (nblk > 256) nblk = 256;
    hipLaunchKernelGGL(fm_w0_reduce1, dim3(nblk), dim3(256), 0, s, a, scratch);
    hipLaunchKernelGGL(fm_w0_reduce2, dim3(1), dim3(256), 0, s, a, scratch, nblk);
    return hipGetLastError();
}

hipError_t fm_launch_w0_apply(const FmArgs &a, hipStream_t s) {
    int64_t blocks = (a.n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fm_w0_apply, dim3((unsigned)blocks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(fm_w0_commit, dim3(1), dim3(1), 0, s, a);
    return hipGetLastError();
}

hipError_t fm_launch_init(const FmArgs &a, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    int64_t blocks = (a.n + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(fm_init_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t fm_launch_predict(const FmArgs &a, int64_t n, const int32_t *tu, const int32_t *tj, const int32_t *tc,
                             int bound, double lo, double hi, double *out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(fm_predict_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, n, tu, tj, tc, bound, lo, hi, out);
    return hipGetLastError();
}

} // namespace cmi

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
// One sweep = 1 (w0) + 3 (w: users, items, contexts) + 3k (V, per factor) phases.
//
// HBM traffic is what bounds a sweep, so the per-rating state is kept to ONE fp64 array:
//   * the reference's cache Q[i][f] = sum_l V[l][f] x_il (FM.java:134-146, 209-210) is not stored (k*size*8 bytes,
//     re-read and re-written by every factor phase): with three features per rating it is V[u][f] + V[item][f] +
//     xc*V[ctx][f], gathered from a dense copy of column f (`col`, p doubles: L2/MALL resident).  Same value up to
//     rounding (the reference accumulates deltas into Q; RMSE holds 1e-9 against the order-exact oracle).
//   * storage order = sorted by user, so the user field streams errors[] sequentially (wave per user, fused
//     reduce + apply);
//   * the item / context fields only GATHER errors[] (their CSR carries the other two feature ids) and leave their
//     coordinate deltas next to the column entries (tab[].y); the next user phase (or the w0 phase of the next sweep) folds them into
//     errors[] on its sequential pass.  No random write ever happens.
// fp64 throughout (the reference's precision); sums are tree-reduced.  Gather/stream work: no MFMA.
//
// Split reduce/apply modes exist so a multi-GPU host can all-reduce (num, den) between them.
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

// true error of storage position i (pending item / context deltas folded in)
__device__ __forceinline__ double fm_err(const FmArgs &a, int64_t i, int jj, int c) {
    double e = a.R[i].x;
    if (a.pending & 1) e += a.tab[a.n_users + jj].y;
    if ((a.pending & 2) && c < a.n_conds) e += a.xc * a.tab[(int64_t)a.n_users + a.n_items + c].y;
    return e;
}

// supporting rating s of coordinate l of FIELD -> true error et and h = x_il * (sum of the OTHER features' column
// entries times their values) = x*Q[i][f] - x*x*theta of FM.java:178,198.  One 16-byte gather per table touched:
// tab[] pairs a coordinate's column entry with its pending delta, R[] pairs a rating's error with its user's entry.
template <int FIELD>
__device__ __forceinline__ void fm_row(const FmArgs &a, int f, int64_t l, int64_t s, double x, double &et, double &h) {
    const int64_t jbase = a.n_users, cbase = (int64_t)a.n_users + a.n_items;
    double other = 0.0;
    if (FIELD == 0) {
        const int jj = a.j[s], c = a.ctx[s];
        const double2 t = a.tab[jbase + jj];
        et = a.R[s].x;
        if (a.pending & 1) et += t.y;
        other = t.x;
        if (c < a.n_conds) {
            const double2 tc = a.tab[cbase + c];
            if (a.pending & 2) et += a.xc * tc.y;
            other += a.xc * tc.x;
        }
    } else if (FIELD == 1) { // pending deltas are always folded before an item phase
        const double2 r = a.R[a.sup[1][s]];
        const int c = a.sup_b[1][s];
        et = r.x;
        other = r.y;
        if (c < a.n_conds) other += a.xc * a.tab[cbase + c].x;
    } else {
        const double2 r = a.R[a.sup[2][s]];
        const double2 t = a.tab[jbase + a.sup_b[2][s]];
        et = r.x;
        if (a.pending & 1) et += t.y;
        other = r.y + t.x;
    }
    h = f < 0 ? x : x * other;
}

// MODE 0: reduce only (writes part[l], part[count+l]); MODE 1: apply only (reads part); MODE 2: fused.
// f < 0: linear weights w; f >= 0: factor column f of V (working copy in a.tab[].x).
// h_i = x_il * (sum of the OTHER features' column entries times their values) = x*Q[i][f] - x*x*theta of FM.java:178,198.
template <int BLOCK, int MODE, int FIELD>
__global__ __launch_bounds__(BLOCK) void fm_field_kernel(FmArgs a, int f) {
    __shared__ double lds[BLOCK / 64 + 2];
    const int l = blockIdx.x; // coordinate within the field
    const int64_t b = a.sup_off[FIELD][l], e = a.sup_off[FIELD][l + 1];
    const int64_t ubase = 0, jbase = a.n_users, cbase = (int64_t)a.n_users + a.n_items;
    const int64_t base = FIELD == 0 ? ubase : (FIELD == 1 ? jbase : cbase);
    const double theta = f < 0 ? a.w[base + l] : a.tab[base + l].x;
    const double x = FIELD == 2 ? a.xc : 1.0;
    double num = 0.0, den = 0.0;
    if (MODE != 1) {
        for (int64_t s = b + threadIdx.x; s < e; s += BLOCK) {
            double et, h;
            fm_row<FIELD>(a, f, l, s, x, et, h);
            num += (et - theta * h) * h;
            den += h * h;
        }
        num = block_sum<BLOCK>(num, lds);
        den = block_sum<BLOCK>(den, lds);
        if (MODE == 0) {
            if (threadIdx.x == 0) {
                a.part[l] = num;
                a.part[a.field_count[FIELD] + l] = den;
            }
            return;
        }
    } else {
        num = a.part[l];
        den = a.part[a.field_count[FIELD] + l];
    }
    const double reg = f < 0 ? a.regLw : a.regLf;
    const double upd = 0.0 - num / (den + (double)a.global_size * reg);
    const double delta = upd - theta;
    if (FIELD == 0) { // the sequential field: errors[] are rewritten here, pending deltas folded in
        for (int64_t s = b + threadIdx.x; s < e; s += BLOCK)
            a.R[s] = make_double2(fm_err(a, s, a.j[s], a.ctx[s]) + delta * x, f < 0 ? 0.0 : upd);
        if (BLOCK > 64) __syncthreads();
    }
    if (threadIdx.x == 0) {
        // errors[i] += delta * x of an item / context coordinate is applied lazily by the next sequential pass
        if (f < 0) {
            a.w[base + l] = upd;
            if (FIELD != 0) a.tab[base + l].y = delta;
        } else {
            a.tab[base + l] = make_double2(upd, FIELD == 0 ? 0.0 : delta);
            a.V[(size_t)(base + l) * a.k + f] = upd;
        }
    }
}

// Short supports (the usual case: a user's or an item's ratings): LANES-wide lane groups, 64/LANES coordinates per
// wave, up to 4*LANES supporting ratings held in registers -- one pass over the support instead of two, and a quarter
// of the waves.  A phase over short supports is bound by the per-wave chain of dependent loads (offsets -> ids /
// errors -> column gathers), not by bytes, so fewer, fuller waves and a shorter chain are what count.  Coordinates
// with more than 4*LANES ratings take the two-loop path inside the same kernel.
template <int LANES, int MODE, int FIELD>
__global__ __launch_bounds__(256) void fm_field_sub(FmArgs a, int f) {
    constexpr int CH = 4;
    const int sub = threadIdx.x % LANES;
    const int64_t l = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LANES; // coordinate within the field
    const bool live = l < a.field_count[FIELD];
    const int64_t b = live ? a.sup_off[FIELD][l] : 0, e = live ? a.sup_off[FIELD][l + 1] : 0;
    const int64_t ubase = 0, jbase = a.n_users, cbase = (int64_t)a.n_users + a.n_items;
    const int64_t base = FIELD == 0 ? ubase : (FIELD == 1 ? jbase : cbase);
    const double theta = !live ? 0.0 : (f < 0 ? a.w[base + l] : a.tab[base + l].x);
    const double x = FIELD == 2 ? a.xc : 1.0;
    const bool small = (e - b) <= CH * LANES;
    double et[CH], num = 0.0, den = 0.0;
    if (MODE != 1) {
        if (small) {
#pragma unroll
            for (int c4 = 0; c4 < CH; ++c4) {
                const int64_t s = b + c4 * LANES + sub;
                et[c4] = 0.0;
                if (s < e) {
                    double h;
                    fm_row<FIELD>(a, f, l, s, x, et[c4], h);
                    num += (et[c4] - theta * h) * h;
                    den += h * h;
                }
            }
        } else {
            for (int64_t s = b + sub; s < e; s += LANES) {
                double e1, h;
                fm_row<FIELD>(a, f, l, s, x, e1, h);
                num += (e1 - theta * h) * h;
                den += h * h;
            }
        }
#pragma unroll
        for (int m = LANES / 2; m >= 1; m >>= 1) {
            num += __shfl_xor(num, m, 64);
            den += __shfl_xor(den, m, 64);
        }
        if (MODE == 0) {
            if (live && sub == 0) {
                a.part[l] = num;
                a.part[a.field_count[FIELD] + l] = den;
            }
            return;
        }
    } else if (live) {
        num = a.part[l];
        den = a.part[a.field_count[FIELD] + l];
    }
    if (!live) return;
    const double reg = f < 0 ? a.regLw : a.regLf;
    const double upd = 0.0 - num / (den + (double)a.global_size * reg);
    const double delta = upd - theta;
    if (FIELD == 0) { // the sequential field rewrites errors[] with the pending deltas folded in
        if (MODE == 2 && small) {
#pragma unroll
            for (int c4 = 0; c4 < CH; ++c4) {
                const int64_t s = b + c4 * LANES + sub;
                if (s < e) a.R[s] = make_double2(et[c4] + delta * x, f < 0 ? 0.0 : upd);
            }
        } else {
            for (int64_t s = b + sub; s < e; s += LANES)
                a.R[s] = make_double2(fm_err(a, s, a.j[s], a.ctx[s]) + delta * x, f < 0 ? 0.0 : upd);
        }
    }
    if (sub == 0) {
        // errors[i] += delta * x of an item / context coordinate is applied lazily by the next sequential pass
        if (f < 0) {
            a.w[base + l] = upd;
            if (FIELD != 0) a.tab[base + l].y = delta;
        } else {
            a.tab[base + l] = make_double2(upd, FIELD == 0 ? 0.0 : delta);
            a.V[(size_t)(base + l) * a.k + f] = upd;
        }
    }
}

__global__ __launch_bounds__(256) void fm_col_load(FmArgs a, int f, int64_t p) {
    for (int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x; l < p; l += (int64_t)gridDim.x * 256)
        a.tab[l].x = a.V[(size_t)l * a.k + f]; // .y (a pending delta of the previous factor's phase) is kept
}

// R[i].y = column entry of the rating's user (only needed when an item / context phase of factor f is driven without
// the user phase of f right before it)
__global__ __launch_bounds__(256) void fm_uval_kernel(FmArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) a.R[i].y = a.tab[a.u[i]].x;
}

// errors[i] += pending deltas (only needed when phases are driven out of the usual order)
__global__ __launch_bounds__(256) void fm_flush_kernel(FmArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256)
        a.R[i].x = fm_err(a, i, a.j[i], a.ctx[i]);
}

// w0 phase, reduce: part[0] = sum(err_i - w0) over the local ratings (fixed two-stage tree)
__global__ __launch_bounds__(256) void fm_w0_reduce1(FmArgs a, double *scratch) {
    __shared__ double lds[6];
    const int64_t chunk = (a.n + gridDim.x - 1) / gridDim.x;
    const int64_t b = (int64_t)blockIdx.x * chunk, e = (b + chunk) < a.n ? (b + chunk) : a.n;
    const double w0 = *a.w0;
    double s = 0.0;
    for (int64_t i = b + threadIdx.x; i < e; i += 256) s += fm_err(a, i, a.j[i], a.ctx[i]) - w0;
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
    // `size + regLw` is int + float in the reference (FM.java:47,161): Java's binary numeric promotion makes it a FLOAT sum (at C4's
    // 25 M ratings regLw vanishes in it) -- found by executing the reference's source, tests/test_reference_src_golden.py
    const double upd = 0.0 - a.part[0] / (double)((float)(int)a.global_size + (float)a.regLw);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256)
        a.R[i].x = fm_err(a, i, a.j[i], a.ctx[i]) + upd - w0;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.part[2] = upd; // committed to *w0 by fm_w0_commit after all blocks read w0
}
__global__ void fm_w0_commit(FmArgs a) { *a.w0 = a.part[2]; }

// pre-pass (FM.java:117-146): errors[i] = r_i - predict(i) (Q is not materialised, see the header).  One wave per rating.
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
            pair += s1 * s1 - s2;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) pair += __shfl_xor(pair, m, 64);
        if (lane == 0) {
            double pred = *a.w0 + a.w[u];
            pred += a.w[a.n_users + j];
            if (has_c) pred += a.w[a.n_users + a.n_items + c] * a.xc;
            a.R[i] = make_double2(a.r[i] - (pred + 0.5 * pair), 0.0);
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

template <int MODE, int FIELD>
static hipError_t launch_field_mode(const FmArgs &a, int f, int64_t avg_support, hipStream_t s) {
    const int count = a.field_count[FIELD];
    if (count <= 0) return hipSuccess;
    if (avg_support > 2048) // few coordinates with very long supports: a whole 1024-thread workgroup each
        hipLaunchKernelGGL((fm_field_kernel<1024, MODE, FIELD>), dim3(count), dim3(1024), 0, s, a, f);
    else if (avg_support > 192)
        hipLaunchKernelGGL((fm_field_kernel<256, MODE, FIELD>), dim3(count), dim3(256), 0, s, a, f);
    else if (avg_support > 48) // 64/LANES coordinates per wave
        hipLaunchKernelGGL((fm_field_sub<32, MODE, FIELD>), dim3((count + 7) / 8), dim3(256), 0, s, a, f);
    else
        hipLaunchKernelGGL((fm_field_sub<16, MODE, FIELD>), dim3((count + 15) / 16), dim3(256), 0, s, a, f);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_field(const FmArgs &a, int field, int f, hipStream_t s) {
    const int64_t avg = a.field_count[field] > 0 ? a.n / a.field_count[field] : 0; // context supports are a subset: fine
    switch (field) {
    case 0: return launch_field_mode<MODE, 0>(a, f, avg, s);
    case 1: return launch_field_mode<MODE, 1>(a, f, avg, s);
    default: return launch_field_mode<MODE, 2>(a, f, avg, s);
    }
}

hipError_t fm_launch_field(const FmArgs &a, int field, int f, int mode, hipStream_t s) {
    switch (mode) {
    case 0: return launch_field<0>(a, field, f, s);
    case 1: return launch_field<1>(a, field, f, s);
    default: return launch_field<2>(a, field, f, s);
    }
}

hipError_t fm_launch_col_load(const FmArgs &a, int f, hipStream_t s) {
    const int64_t p = (int64_t)a.n_users + a.n_items + a.n_conds;
    int64_t blocks = (p + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fm_col_load, dim3((unsigned)blocks), dim3(256), 0, s, a, f, p);
    return hipGetLastError();
}

hipError_t fm_launch_uval(const FmArgs &a, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    int64_t blocks = (a.n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(fm_uval_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t fm_launch_flush(const FmArgs &a, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    int64_t blocks = (a.n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(fm_flush_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
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

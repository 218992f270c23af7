// ext_kernels.hip -- SVD++ and the similarity-based CAMF family (CAMF_ICS / CAMF_LCS / CAMF_MCS) on gfx950.
//
// Reference loops: src/carskit/alg/baseline/cf/SVDPlusPlus.java:58-128 and
// src/carskit/alg/cars/adaptation/dependent/sim/CAMF_ICS.java:62-131, CAMF_LCS.java:63-146, CAMF_MCS.java:70-165.
// In these models EVERY rating updates parameters that every other rating reads -- the condition-similarity scalars / vectors /
// positions of the CAMF_*CS family, and in SVD++ the implicit-feedback rows Y[k] of all items its user rated -- so no two
// tuples commute and the exact semantics are ONE dependent chain in CRS order (CMI_FLAG_SCHED_SERIAL is required, as for
// CAMF_C).  Two kernels:
//   ext_serial_strict<T, MODEL>  one lane performs the reference's operations one by one, in its order, one rounding per
//                                operator (-ffp-contract=off): with T = double the model AND the epoch loss are bit-identical to
//                                the Java arithmetic.  The correctness anchor.
//   ext_serial_wave<T, MODEL>    one wave64, lane f owns factors f, f+64, ...: row traffic is coalesced, dot products and the
//                                loss are tree sums (DPP), the scalar similarity chain runs on lane-uniform values.  Same
//                                per-element expressions; results within rounding (1e-9 in fp64).
// A GPU is the wrong machine for a strictly sequential chain: these exist so that the recommender names resolve to
// the same library with the same parity guarantees, not for speed.
#include "mf_sgd_kernels.hpp"
#include "sgd_device.hpp"

#include <cmath>

namespace cmi {

// ---------------------------------------------------------------------------------------------
// strict: one lane, the reference's operation sequence
// ---------------------------------------------------------------------------------------------

template <typename T>
__device__ __forceinline__ T seq_dot(const T *a, const T *b, int k) { // librec DenseMatrix.rowMult: s = 0; s += a[f]*b[f]
    T s = 0;
    for (int f = 0; f < k; ++f) s += a[f] * b[f];
    return s;
}

template <typename T>
__device__ __forceinline__ T sym_get(const ExtArgs<T> &a, int x, int y) { return a.cc[(size_t)x * a.n_conds + y]; }
template <typename T>
__device__ __forceinline__ void sym_set(const ExtArgs<T> &a, int x, int y, T v) { // librec SymmMatrix: one cell for (x,y) and (y,x)
    a.cc[(size_t)x * a.n_conds + y] = v;
    a.cc[(size_t)y * a.n_conds + x] = v;
}

// predict(u, j, c) in the reference's operation order (SVDPlusPlus.java:138-146, CAMF_ICS.java:53-59, CAMF_LCS.java:43-61,
// CAMF_MCS.java:53-68); conds = the tuple's condition ids (-1 padded to dmax)
template <typename T, int MODEL>
__device__ T ext_predict_seq(const ExtArgs<T> &a, T gm, int uu, int jj, const int32_t *conds) {
    const int k = a.k;
    const T *pu = a.P + (size_t)uu * k, *qj = a.Q + (size_t)jj * k;
    if (MODEL == SVDPP) {
        T pred = gm + a.userBias[uu] + a.itemBias[jj] + seq_dot(pu, qj, k);
        const int32_t b = a.ui_ptr[uu], e = a.ui_ptr[uu + 1];
        const T w = (T)sqrt((double)(e - b));
        for (int32_t q = b; q < e; ++q) pred += seq_dot(a.Y + (size_t)a.ui_items[q] * k, qj, k) / w;
        return pred;
    }
    T pred = seq_dot(pu, qj, k);
    if (MODEL == CAMF_MCS) {
        T dist = 0;
        for (int i = 0; i < a.dmax && conds[i] >= 0 && i < a.n_empty; ++i) {
            const T d = a.cv[conds[i]] - a.cv[a.empty_conds[i]];
            dist += d * d;
        }
        dist = (T)sqrt((double)dist);
        return pred * ((T)1 - dist);
    }
    for (int i = 0; i < a.dmax && conds[i] >= 0 && i < a.n_empty; ++i) {
        const int c1 = conds[i], c2 = a.empty_conds[i];
        pred = pred * (MODEL == CAMF_ICS ? sym_get(a, c1, c2) : seq_dot(a.cf + (size_t)c1 * a.num_f, a.cf + (size_t)c2 * a.num_f, a.num_f));
    }
    return pred;
}

constexpr int EXT_MAX_DIMS = 16;

template <typename T, int MODEL>
__global__ __launch_bounds__(64) void ext_serial_strict(ExtArgs<T> a, int64_t n, double *loss_out) {
    if (threadIdx.x != 0) return;
    const HParams hp = *a.hp;
    const T lr = (T)hp.lr, regU = (T)hp.regU, regI = (T)hp.regI, regB = (T)hp.regB, regC = (T)hp.regC, gm = (T)hp.gm;
    const int k = a.k;
    double loss = 0.0; // the reference's `loss` is a double; with T = float the terms are rounded to float first
    for (int64_t t = 0; t < n; ++t) {
        const int uu = a.su[t], jj = a.sj[t];
        const T rr = a.sr[t];
        T *pu = a.P + (size_t)uu * k, *qj = a.Q + (size_t)jj * k;
        const int32_t *conds = a.sconds + t * a.dmax;
        if (MODEL == SVDPP) {
            const T pred = ext_predict_seq<T, MODEL>(a, gm, uu, jj, conds);
            const T e = rr - pred;
            loss += (double)(e * e);
            const int32_t b = a.ui_ptr[uu], en = a.ui_ptr[uu + 1];
            const T w = (T)sqrt((double)(en - b));
            const T bu = a.userBias[uu];
            a.userBias[uu] = bu + lr * (e - regB * bu);
            loss += (double)((regB * bu) * bu);
            const T bj = a.itemBias[jj];
            a.itemBias[jj] = bj + lr * (e - regB * bj);
            loss += (double)((regB * bj) * bj);
            for (int f = 0; f < k; ++f) { // sum_ys[f] is needed before any Y moves; Y[.,f] only moves in iteration f
                T sum_f = 0;
                for (int32_t q = b; q < en; ++q) sum_f += a.Y[(size_t)a.ui_items[q] * k + f];
                const T sum_ys = w > (T)0 ? sum_f / w : sum_f;
                const T puf = pu[f], qjf = qj[f];
                pu[f] = puf + lr * (e * qjf - regU * puf);
                qj[f] = qjf + lr * (e * (puf + sum_ys) - regI * qjf);
                loss += (double)((regU * puf) * puf + (regI * qjf) * qjf);
                for (int32_t q = b; q < en; ++q) {
                    T *y = a.Y + (size_t)a.ui_items[q] * k + f;
                    const T ykf = *y;
                    *y = ykf + lr * ((e * qjf) / w - regU * ykf);
                    loss += (double)((regU * ykf) * ykf);
                }
            }
            continue;
        }
        int i1[EXT_MAX_DIMS], i2[EXT_MAX_DIMS];
        T val[EXT_MAX_DIMS];
        int nupd = 0;
        const T dot = seq_dot(pu, qj, k);
        T pred = dot, scale;
        T e;
        if (MODEL == CAMF_MCS) {
            T dist = 0;
            for (int i = 0; i < a.dmax && conds[i] >= 0 && i < a.n_empty; ++i) {
                const int c1 = conds[i], c2 = a.empty_conds[i];
                const T pos1 = a.cv[c1], pos2 = a.cv[c2];
                const T diff = pos1 - pos2;
                dist += diff * diff;
                if (c1 != c2) {
                    i1[nupd] = c1, i2[nupd] = c2, val[nupd] = diff;
                    ++nupd;
                }
                loss += (double)((regC * pos1) * pos1 + (regC * pos2) * pos2);
            }
            dist = (T)sqrt((double)dist);
            pred *= (T)1 - dist;
            e = rr - pred;
            loss += (double)(e * e);
            for (int q = 0; q < nupd; ++q) {
                const T pos1 = a.cv[i1[q]], pos2 = a.cv[i2[q]];
                if (dist == (T)0) dist = (T)a.lowbound; // sticks for the rest of the tuple (CAMF_MCS.java:121-122)
                T p1 = pos1 + lr * (((e * dot) * val[q]) / dist - regC * pos1);
                T p2 = pos2 - lr * (((e * dot) * val[q]) / dist + regC * pos2);
                p1 = p1 < (T)0 ? (T)a.lowbound : p1;
                p1 = p1 > (T)a.upbound ? (T)a.upbound - (T)a.lowbound : p1;
                p2 = p2 < (T)0 ? (T)a.lowbound : p2;
                p2 = p2 > (T)a.upbound ? (T)a.upbound - (T)a.lowbound : p2;
                a.cv[i1[q]] = p1;
                a.cv[i2[q]] = p2;
            }
            scale = (T)1 - dist;
        } else {
            T simc = 1;
            for (int i = 0; i < a.dmax && conds[i] >= 0 && i < a.n_empty; ++i) {
                const int c1 = conds[i], c2 = a.empty_conds[i];
                T sim = 1;
                if (c1 != c2) {
                    sim = MODEL == CAMF_ICS ? sym_get(a, c1, c2) : seq_dot(a.cf + (size_t)c1 * a.num_f, a.cf + (size_t)c2 * a.num_f, a.num_f);
                    i1[nupd] = c1, i2[nupd] = c2, val[nupd] = sim;
                    ++nupd;
                    simc *= sim;
                }
                if (MODEL == CAMF_ICS) loss += (double)((regC * sim) * sim);
                pred = pred * sim;
            }
            e = rr - pred;
            loss += (double)(e * e);
            for (int q = 0; q < nupd; ++q) {
                if (MODEL == CAMF_ICS) {
                    T upd = val[q];
                    upd += lr * (((e * dot) * simc) / upd - regC * upd);
                    sym_set(a, i1[q], i2[q], upd);
                } else {
                    T *c1 = a.cf + (size_t)i1[q] * a.num_f, *c2 = a.cf + (size_t)i2[q] * a.num_f;
                    const T sim = val[q];
                    for (int f = 0; f < a.num_f; ++f) {
                        const T c1f = c1[f], c2f = c2[f];
                        c1[f] = c1f + lr * ((((e * dot) * simc) * c2f) / sim - regC * c1f);
                        c2[f] = c2f + lr * ((((e * dot) * simc) * c1f) / sim - regC * c2f);
                        loss += (double)((regC * c1f) * c1f + (regC * c2f) * c2f);
                    }
                }
            }
            scale = simc;
        }
        for (int f = 0; f < k; ++f) {
            const T puf = pu[f], qjf = qj[f];
            pu[f] = puf + lr * ((e * qjf) * scale - regU * puf);
            qj[f] = qjf + lr * ((e * puf) * scale - regI * qjf);
            loss += (double)((regU * puf) * puf + (regI * qjf) * qjf);
        }
    }
    loss_out[0] = loss * (MODEL == CAMF_MCS ? 0.05 : 0.5); // CAMF_MCS.java:158 really scales by 0.05
}

// ---------------------------------------------------------------------------------------------
// wave: lane f owns factors f, f+64, ...; tree sums
// ---------------------------------------------------------------------------------------------

template <typename T>
__device__ __forceinline__ T wave_dot(const T *a, const T *b, int k, int lane) {
    T part = 0;
    for (int f = lane; f < k; f += 64) part += a[f] * b[f];
    return wave_sum64(part);
}

template <typename T, int MODEL>
__global__ __launch_bounds__(64) void ext_serial_wave(ExtArgs<T> a, int64_t n, double *loss_out) {
    const int lane = threadIdx.x;
    const HParams hp = *a.hp;
    const T lr = (T)hp.lr, regU = (T)hp.regU, regI = (T)hp.regI, regB = (T)hp.regB, regC = (T)hp.regC, gm = (T)hp.gm;
    const int k = a.k;
    double loss = 0.0; // lane-uniform scalar terms
    double lpart = 0.0; // per-lane factor terms, tree-reduced once at the end
    for (int64_t t = 0; t < n; ++t) {
        const int uu = a.su[t], jj = a.sj[t];
        const T rr = a.sr[t];
        T *pu = a.P + (size_t)uu * k, *qj = a.Q + (size_t)jj * k;
        const int32_t *conds = a.sconds + t * a.dmax;
        const T dot = wave_dot(pu, qj, k, lane);
        T e, scale = 1;
        if (MODEL == SVDPP) {
            const int32_t b = a.ui_ptr[uu], en = a.ui_ptr[uu + 1];
            const T w = (T)sqrt((double)(en - b));
            const T bu = a.userBias[uu], bj = a.itemBias[jj];
            T pred = gm + bu + bj + dot;
            for (int32_t q = b; q < en; ++q) pred += wave_dot(a.Y + (size_t)a.ui_items[q] * k, qj, k, lane) / w;
            e = rr - pred;
            loss += (double)(e * e) + (double)((regB * bu) * bu) + (double)((regB * bj) * bj);
            if (lane == 0) {
                a.userBias[uu] = bu + lr * (e - regB * bu);
                a.itemBias[jj] = bj + lr * (e - regB * bj);
            }
            for (int f = lane; f < k; f += 64) {
                T sum_f = 0;
                for (int32_t q = b; q < en; ++q) sum_f += a.Y[(size_t)a.ui_items[q] * k + f];
                const T sum_ys = w > (T)0 ? sum_f / w : sum_f;
                const T puf = pu[f], qjf = qj[f];
                pu[f] = puf + lr * (e * qjf - regU * puf);
                qj[f] = qjf + lr * (e * (puf + sum_ys) - regI * qjf);
                lpart += (double)((regU * puf) * puf + (regI * qjf) * qjf);
                for (int32_t q = b; q < en; ++q) {
                    T *y = a.Y + (size_t)a.ui_items[q] * k + f;
                    const T ykf = *y;
                    *y = ykf + lr * ((e * qjf) / w - regU * ykf);
                    lpart += (double)((regU * ykf) * ykf);
                }
            }
            continue;
        }
        // the similarity chain: every lane evaluates the same scalar expressions on the same values (loads are wave-uniform)
        int i1[EXT_MAX_DIMS], i2[EXT_MAX_DIMS];
        T val[EXT_MAX_DIMS];
        int nupd = 0;
        T pred = dot;
        if (MODEL == CAMF_MCS) {
            T dist = 0;
            for (int i = 0; i < a.dmax && conds[i] >= 0 && i < a.n_empty; ++i) {
                const int c1 = conds[i], c2 = a.empty_conds[i];
                const T pos1 = a.cv[c1], pos2 = a.cv[c2];
                const T diff = pos1 - pos2;
                dist += diff * diff;
                if (c1 != c2) {
                    i1[nupd] = c1, i2[nupd] = c2, val[nupd] = diff;
                    ++nupd;
                }
                loss += (double)((regC * pos1) * pos1 + (regC * pos2) * pos2);
            }
            dist = (T)sqrt((double)dist);
            pred *= (T)1 - dist;
            e = rr - pred;
            loss += (double)(e * e);
            for (int q = 0; q < nupd; ++q) {
                const T pos1 = a.cv[i1[q]], pos2 = a.cv[i2[q]];
                if (dist == (T)0) dist = (T)a.lowbound;
                T p1 = pos1 + lr * (((e * dot) * val[q]) / dist - regC * pos1);
                T p2 = pos2 - lr * (((e * dot) * val[q]) / dist + regC * pos2);
                p1 = p1 < (T)0 ? (T)a.lowbound : p1;
                p1 = p1 > (T)a.upbound ? (T)a.upbound - (T)a.lowbound : p1;
                p2 = p2 < (T)0 ? (T)a.lowbound : p2;
                p2 = p2 > (T)a.upbound ? (T)a.upbound - (T)a.lowbound : p2;
                if (lane == 0) {
                    a.cv[i1[q]] = p1;
                    a.cv[i2[q]] = p2;
                }
            }
            scale = (T)1 - dist;
        } else {
            T simc = 1;
            for (int i = 0; i < a.dmax && conds[i] >= 0 && i < a.n_empty; ++i) {
                const int c1 = conds[i], c2 = a.empty_conds[i];
                T sim = 1;
                if (c1 != c2) {
                    sim = MODEL == CAMF_ICS ? sym_get(a, c1, c2) : wave_dot(a.cf + (size_t)c1 * a.num_f, a.cf + (size_t)c2 * a.num_f, a.num_f, lane);
                    i1[nupd] = c1, i2[nupd] = c2, val[nupd] = sim;
                    ++nupd;
                    simc *= sim;
                }
                if (MODEL == CAMF_ICS) loss += (double)((regC * sim) * sim);
                pred = pred * sim;
            }
            e = rr - pred;
            loss += (double)(e * e);
            for (int q = 0; q < nupd; ++q) {
                if (MODEL == CAMF_ICS) {
                    T upd = val[q];
                    upd += lr * (((e * dot) * simc) / upd - regC * upd);
                    if (lane == 0) sym_set(a, i1[q], i2[q], upd);
                } else {
                    T *c1 = a.cf + (size_t)i1[q] * a.num_f, *c2 = a.cf + (size_t)i2[q] * a.num_f;
                    const T sim = val[q];
                    for (int f = lane; f < a.num_f; f += 64) {
                        const T c1f = c1[f], c2f = c2[f];
                        c1[f] = c1f + lr * ((((e * dot) * simc) * c2f) / sim - regC * c1f);
                        c2[f] = c2f + lr * ((((e * dot) * simc) * c1f) / sim - regC * c2f);
                        lpart += (double)((regC * c1f) * c1f + (regC * c2f) * c2f);
                    }
                }
            }
            scale = simc;
        }
        for (int f = lane; f < k; f += 64) {
            const T puf = pu[f], qjf = qj[f];
            pu[f] = puf + lr * ((e * qjf) * scale - regU * puf);
            qj[f] = qjf + lr * ((e * puf) * scale - regI * qjf);
            lpart += (double)((regU * puf) * puf + (regI * qjf) * qjf);
        }
        // the next tuple may read what other lanes just wrote (rows, similarity cells): same-wave stores and later loads of one
        // address stay ordered in the vector memory pipeline, but the compiler must not hoist the loads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const double total = loss + wave_sum64(lpart);
    if (lane == 0) loss_out[0] = total * (MODEL == CAMF_MCS ? 0.05 : 0.5);
}

// ---------------------------------------------------------------------------------------------
// predict / evalRatings for these models: one wave per tuple, fp64 arithmetic over the stored state
// ---------------------------------------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(256) void ext_eval_kernel(ExtEvalArgs<T> a, int64_t n) {
    __shared__ double s_part[4][5];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * 4;
    double s_abs = 0, s_sq = 0, s_rabs = 0, s_rsq = 0, s_cnt = 0;
    const int k = a.k;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n; t += stride) {
        const int uu = a.u[t], jj = a.j[t];
        const T *pu = a.P + (size_t)uu * k, *qj = a.Q + (size_t)jj * k;
        double part = 0.0;
        for (int f = lane; f < k; f += 64) part += (double)pu[f] * (double)qj[f];
        double pred = wave_sum64(part);
        if (a.model == SVDPP) {
            const int32_t b = a.ui_ptr[uu], e = a.ui_ptr[uu + 1];
            const double w = sqrt((double)(e - b));
            double yq = 0.0;
            for (int32_t q = b; q < e; ++q) {
                const T *y = a.Y + (size_t)a.ui_items[q] * k;
                for (int f = lane; f < k; f += 64) yq += (double)y[f] * (double)qj[f];
            }
            pred = ((a.gm + (double)a.userBias[uu]) + (double)a.itemBias[jj]) + pred + (e > b ? wave_sum64(yq) / w : 0.0);
        } else {
            const int c = a.ctx[t];
            const int32_t b = a.ctx_ptr[c], e = a.ctx_ptr[c + 1];
            double dist = 0.0;
            for (int32_t q = b; q < e && q - b < a.n_empty; ++q) {
                const int c1 = a.ctx_conds[q], c2 = a.empty_conds[q - b];
                if (a.model == CAMF_ICS) pred *= (double)a.cc[(size_t)c1 * a.n_conds + c2];
                else if (a.model == CAMF_LCS) {
                    double sp = 0.0;
                    for (int f = lane; f < a.num_f; f += 64) sp += (double)a.cf[(size_t)c1 * a.num_f + f] * (double)a.cf[(size_t)c2 * a.num_f + f];
                    pred *= wave_sum64(sp);
                } else {
                    const double d = (double)a.cv[c1] - (double)a.cv[c2];
                    dist += d * d;
                }
            }
            if (a.model == CAMF_MCS) pred *= 1.0 - sqrt(dist);
        }
        if (a.bound) {
            if (pred > a.hi) pred = a.hi;
            if (pred < a.lo) pred = a.lo;
        }
        if (a.preds && lane == 0) a.preds[t] = pred;
        if (a.r && !isnan(pred)) {
            const double rate = a.r[t];
            const double rpred = floor(pred / a.min_rate + 0.5) * a.min_rate;
            const double err = fabs(rate - pred), rerr = fabs(rate - rpred);
            s_abs += err;
            s_sq += err * err;
            s_rabs += rerr;
            s_rsq += rerr * rerr;
            s_cnt += 1.0;
        }
    }
    if (a.part) {
        if (lane == 0) {
            s_part[wave][0] = s_abs;
            s_part[wave][1] = s_sq;
            s_part[wave][2] = s_rabs;
            s_part[wave][3] = s_rsq;
            s_part[wave][4] = s_cnt;
        }
        __syncthreads();
        if (threadIdx.x < 5) {
            const int c = threadIdx.x;
            a.part[(size_t)blockIdx.x * 5 + c] = ((s_part[0][c] + s_part[1][c]) + s_part[2][c]) + s_part[3][c];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// operands of the ranking evaluation (rank_kernels.hip: score(q, j) = <a_q, b_j> + const_q)
//   CAMF_ICS / LCS / MCS : predict = <P_u,Q_j> * s(c) with s the product of the context's similarities (or 1 - dist):
//                          a_q = s(c) * P_u, b_j = Q_j, const = 0
//   SVD++                : predict = gm + bu + bj + <Q_j, P_u + (sum_{k in N(u)} Y_k) / w>:
//                          a_q = [P_u + ysum_u / w | 1], b_j = [Q_j | itemBias_j], const = gm + bu
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void ext_rank_items(ExtEvalArgs<T> a, const int32_t *cand, T *B, int kp) { // one block per candidate item
    const int j = cand[blockIdx.x];
    T *dst = B + (size_t)blockIdx.x * kp;
    for (int f = threadIdx.x; f < kp; f += blockDim.x) {
        T v = 0;
        if (f < a.k) v = a.Q[(size_t)j * a.k + f];
        else if (f == a.k && a.model == SVDPP) v = a.itemBias[j];
        dst[f] = v;
    }
}

template <typename T>
__global__ void ext_rank_queries(ExtEvalArgs<T> a, const int32_t *qu, const int32_t *qc, T *A, T *row_const, int kp) { // one block per query
    const int u = qu[blockIdx.x], c = qc[blockIdx.x];
    T *dst = A + (size_t)blockIdx.x * kp;
    double scale = 1.0;
    if (a.model != SVDPP) { // every thread evaluates the same few scalars
        double dist = 0.0;
        const int32_t b = a.ctx_ptr[c], e = a.ctx_ptr[c + 1];
        for (int32_t q = b; q < e && q - b < a.n_empty; ++q) {
            const int c1 = a.ctx_conds[q], c2 = a.empty_conds[q - b];
            if (a.model == CAMF_ICS) scale *= (double)a.cc[(size_t)c1 * a.n_conds + c2];
            else if (a.model == CAMF_LCS) {
                double sp = 0.0;
                for (int f = 0; f < a.num_f; ++f) sp += (double)a.cf[(size_t)c1 * a.num_f + f] * (double)a.cf[(size_t)c2 * a.num_f + f];
                scale *= sp;
            } else {
                const double d = (double)a.cv[c1] - (double)a.cv[c2];
                dist += d * d;
            }
        }
        if (a.model == CAMF_MCS) scale = 1.0 - sqrt(dist);
    }
    const int32_t ib = a.model == SVDPP ? a.ui_ptr[u] : 0, ie = a.model == SVDPP ? a.ui_ptr[u + 1] : 0;
    const double w = sqrt((double)(ie - ib));
    for (int f = threadIdx.x; f < kp; f += blockDim.x) {
        double v = 0.0;
        if (f < a.k) {
            v = (double)a.P[(size_t)u * a.k + f];
            if (a.model == SVDPP) {
                double ys = 0.0;
                for (int32_t q = ib; q < ie; ++q) ys += (double)a.Y[(size_t)a.ui_items[q] * a.k + f];
                if (ie > ib) v += ys / w;
            } else {
                v *= scale;
            }
        } else if (f == a.k && a.model == SVDPP) {
            v = 1.0;
        }
        dst[f] = (T)v;
    }
    if (threadIdx.x == 0) row_const[blockIdx.x] = a.model == SVDPP ? (T)(a.gm + (double)a.userBias[u]) : (T)0;
}

template <typename T>
hipError_t launch_ext_rank_items(const ExtEvalArgs<T> &a, const int32_t *cand, int nc, T *B, int kp, hipStream_t s) {
    if (nc <= 0) return hipSuccess;
    hipLaunchKernelGGL(ext_rank_items<T>, dim3(nc), dim3(128), 0, s, a, cand, B, kp);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_ext_rank_queries(const ExtEvalArgs<T> &a, const int32_t *qu, const int32_t *qc, int nq, T *A, T *row_const, int kp,
                                   hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(ext_rank_queries<T>, dim3(nq), dim3(128), 0, s, a, qu, qc, A, row_const, kp);
    return hipGetLastError();
}
template hipError_t launch_ext_rank_items<float>(const ExtEvalArgs<float> &, const int32_t *, int, float *, int, hipStream_t);
template hipError_t launch_ext_rank_items<double>(const ExtEvalArgs<double> &, const int32_t *, int, double *, int, hipStream_t);
template hipError_t launch_ext_rank_queries<float>(const ExtEvalArgs<float> &, const int32_t *, const int32_t *, int, float *, float *, int, hipStream_t);
template hipError_t launch_ext_rank_queries<double>(const ExtEvalArgs<double> &, const int32_t *, const int32_t *, int, double *, double *, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------

template <typename T, int MODEL>
static hipError_t launch_ext_model(const ExtArgs<T> &a, bool strict, int64_t n, double *loss_out, hipStream_t s) {
    if (strict) hipLaunchKernelGGL((ext_serial_strict<T, MODEL>), dim3(1), dim3(64), 0, s, a, n, loss_out);
    else if (MODEL == SVDPP && svdpp_team_supported(a.k)) return launch_svdpp_team<T>(a, n, loss_out, s);
    else hipLaunchKernelGGL((ext_serial_wave<T, MODEL>), dim3(1), dim3(64), 0, s, a, n, loss_out);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_ext_serial(const ExtArgs<T> &a, int model, bool strict, int64_t n, double *loss_out, hipStream_t s) {
    switch (model) {
    case SVDPP: return launch_ext_model<T, SVDPP>(a, strict, n, loss_out, s);
    case CAMF_ICS: return launch_ext_model<T, CAMF_ICS>(a, strict, n, loss_out, s);
    case CAMF_LCS: return launch_ext_model<T, CAMF_LCS>(a, strict, n, loss_out, s);
    case CAMF_MCS: return launch_ext_model<T, CAMF_MCS>(a, strict, n, loss_out, s);
    }
    return hipErrorInvalidValue;
}
template hipError_t launch_ext_serial<float>(const ExtArgs<float> &, int, bool, int64_t, double *, hipStream_t);
template hipError_t launch_ext_serial<double>(const ExtArgs<double> &, int, bool, int64_t, double *, hipStream_t);

template <typename T>
hipError_t launch_ext_eval(const ExtEvalArgs<T> &a, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(ext_eval_kernel<T>, dim3(eval_blocks(n)), dim3(256), 0, s, a, n);
    return hipGetLastError();
}
template hipError_t launch_ext_eval<float>(const ExtEvalArgs<float> &, int64_t, hipStream_t);
template hipError_t launch_ext_eval<double>(const ExtEvalArgs<double> &, int64_t, hipStream_t);

} // namespace cmi

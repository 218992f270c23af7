// mf_sgd_kernels.hip -- hand-written gfx950 (MI355X, CDNA4) kernels for the CARSKit SGD inner loop.
//
// What one update is (reference: src/carskit/alg/cars/adaptation/dependent/dev/CAMF_CI.java:79-123
// and its siblings CAMF_C/CAMF_CU/CAMF_CUCI, src/carskit/alg/baseline/cf/BiasedMF.java:62-98):
//     pred = gm [+ bu] [+ bj] + <P[u],Q[j]> + sum_c contextBias(c)      e = r - pred
//     bias  += lr * (e - reg * bias)            for every bias entry the model touches
//     P[u]  += lr * (e * Q[j] - regU * P[u])    Q[j] += lr * (e * P[u] - regI * Q[j])   (old values)
// i.e. a gather of two k-vectors and a handful of scalars, a dot product, and an AXPY scatter:
// ~10k flop against ~16k bytes.  HBM-bound, no dense contraction -> no MFMA; the work is coalesced
// 16-byte-per-lane row traffic, DPP row reductions for the dot, and keeping enough tuples in flight.
//
// Three kernel families:
//   sgd_level_fast_f32  fp32 state, k in {64,128,256}: 16 lanes per tuple (4 tuples per wave64), each lane
//                       owns k/16 factors as float4s -> every row load/store instruction moves whole 256-B
//                       segments; the dot is reduced inside a DPP row (row_ror 8/4/2/1, pure VALU).
//   sgd_level_generic   any k / fp64 state / strict order: one wave64 per tuple, lane-strided factors.
//   sgd_serial          one wave64 walks every tuple in the reference's order (exact for all models).
// A "level" is a set of tuples that share no user and no item; their updates commute exactly, so
// running levels back-to-back reproduces the reference's sequential result (see level_schedule.cpp).
//
// Built with -ffp-contract=off: the JVM never fuses a*b+c, and the strict fp64 path is bit-compared
// with the CPU oracle.
#include "mf_sgd_kernels.hpp"
#include "env_knobs.hpp"
#include "sgd_device.hpp"

#include <cstdlib>
#include <cstring>

namespace cmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Device-coherent 16-byte row traffic through raw buffer instructions with cache policy sc0|sc1 (aux 17):
// compiler-tracked (its own s_waitcnt), unlike inline-asm loads whose destination registers the allocator may
// copy before a hand-placed wait.  A raw buffer addresses base + 32-bit byte offset: the table must be < 4 GiB
// (checked on the host; larger models use the level schedule).
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
#define CMI_CPOL_SC0_SC1 17
__device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0xffffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 ld_row_coherent(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, CMI_CPOL_SC0_SC1));
}
__device__ __forceinline__ void st_row_coherent(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)byte_off, 0, CMI_CPOL_SC0_SC1);
}
__device__ __forceinline__ float ld_f32_coherent(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_f32_coherent(float *p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// ---------------------------------------------------------------------------------------------
// fast path: fp32 state, K = 64*VPL, 16 lanes per tuple
// ---------------------------------------------------------------------------------------------

// TPG = tuples each 16-lane group processes CONCURRENTLY (all their loads are issued before the first use):
// more bytes in flight per wave and 1/TPG as many workgroups to dispatch per level.
// ids of one tuple as a lane group sees them (lane l additionally holds the tuple's l-th condition)
struct TupleIds {
    int uu, jj, cond;
    float rr;
    bool live;
};
template <bool HAS_CTX>
__device__ __forceinline__ TupleIds load_tuple_ids(const SgdArgs<float> &a, int64_t begin, int count, int g, int lane_in_group) {
    TupleIds t{0, 0, -1, 0.f, g < count};
    if (t.live) {
        const int64_t s = begin + g;
        t.uu = a.su[s];
        t.jj = a.sj[s];
        t.rr = a.sr[s];
        if (HAS_CTX && lane_in_group < a.dmax) t.cond = a.sconds[s * a.dmax + lane_in_group];
    }
    return t;
}

// The update of TPG tuples by one 16-lane group: tuples g0, g0 + GS, ... of the level [begin, begin + count).
// Returns the group's loss contribution (valid in lane 0 of the group).  GS = groups that work side by side (16 per
// 256-thread workgroup of a level launch; 64 in the single 1024-thread workgroup of a narrow-run launch), so that
// neighbouring groups read neighbouring tuples of the stream.
template <int MODEL, int VPL, int TPG, bool RAGGED, bool COH, int GS, bool PRE = false>
__device__ __forceinline__ double fast_tuples_f32(const SgdArgs<float> &a, int64_t begin, int count, int g0, int l16,
                                                  const TupleIds *pre = nullptr) {
    using M = Traits<MODEL>;
    static_assert(MODEL != CAMF_C, "CAMF_C has no level schedule (shared condBias)");
    // RAGGED: any k with k % 4 == 0 and 64*(VPL-1) < k <= 64*VPL (rows stay 16-byte aligned); float4 slots past k are masked
    const int K = RAGGED ? a.k : 64 * VPL;
    double gloss = 0.0;

    bool live[TPG];
    int uu[TPG], jj[TPG], cond[TPG];
    float rr[TPG];
#pragma unroll
    for (int i = 0; i < TPG; ++i) {
        // PRE (narrow-run launches, TPG = 1): the ids were loaded while the previous level was still being computed
        const TupleIds t = PRE ? *pre : load_tuple_ids<Traits<MODEL>::has_ctx>(a, begin, count, g0 + GS * i, l16);
        live[i] = t.live;
        uu[i] = t.uu;
        jj[i] = t.jj;
        rr[i] = t.rr;
        cond[i] = t.cond;
    }

    // COH (experiment CMI_LEVEL_COHERENT=1): all model traffic device-coherent (sc0 sc1: write-through stores, L2-bypassing
    // loads), so a launch leaves no dirty lines in the XCD L2s for the end-of-kernel write-back
    const __amdgpu_buffer_rsrc_t rsP = table_rsrc(a.P), rsQ = table_rsrc(a.Q);
    float4 *prow[TPG], *qrow[TPG];
    float4 p[TPG][VPL], q[TPG][VPL];
    float bu[TPG], bj[TPG], bic[TPG], buc[TPG];
    float *pic[TPG], *puc[TPG];
#pragma unroll
    for (int i = 0; i < TPG; ++i) {
        prow[i] = reinterpret_cast<float4 *>(a.P + (size_t)uu[i] * K) + l16;
        qrow[i] = reinterpret_cast<float4 *>(a.Q + (size_t)jj[i] * K) + l16;
        bu[i] = bj[i] = bic[i] = buc[i] = 0.f;
        pic[i] = puc[i] = nullptr;
        if (live[i]) {
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                p[i][v] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!RAGGED || 4 * l16 + 64 * v < K) {
                    if (COH) {
                        const f32x4 t = ld_row_coherent(rsP, (uint32_t)(((size_t)uu[i] * K + 4 * l16 + 64 * v) * 4));
                        p[i][v] = make_float4(t.x, t.y, t.z, t.w);
                    } else {
                        p[i][v] = prow[i][v * 16];
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                q[i][v] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!RAGGED || 4 * l16 + 64 * v < K) {
                    if (COH) {
                        const f32x4 t = ld_row_coherent(rsQ, (uint32_t)(((size_t)jj[i] * K + 4 * l16 + 64 * v) * 4));
                        q[i][v] = make_float4(t.x, t.y, t.z, t.w);
                    } else {
                        q[i][v] = qrow[i][v * 16];
                    }
                }
            }
            if (M::has_bu) bu[i] = COH ? ld_f32_coherent(a.userBias + uu[i]) : a.userBias[uu[i]];
            if (M::has_bj) bj[i] = COH ? ld_f32_coherent(a.itemBias + jj[i]) : a.itemBias[jj[i]];
            if (cond[i] >= 0) {
                if (M::has_ic) {
                    pic[i] = a.icBias + (size_t)jj[i] * a.n_conds + cond[i];
                    bic[i] = COH ? ld_f32_coherent(pic[i]) : *pic[i];
                }
                if (M::has_uc) {
                    puc[i] = a.ucBias + (size_t)uu[i] * a.n_conds + cond[i];
                    buc[i] = COH ? ld_f32_coherent(puc[i]) : *puc[i];
                }
            }
        }
    }

    const HParams hp = *a.hp;
    const float lr = (float)hp.lr, regU = (float)hp.regU, regI = (float)hp.regI, regB = (float)hp.regB,
                regC = (float)hp.regC, gm = (float)hp.gm;

#pragma unroll
    for (int i = 0; i < TPG; ++i) {
        if (!live[i]) continue; // group-uniform
        float part = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            part += p[i][v].x * q[i][v].x;
            part += p[i][v].y * q[i][v].y;
            part += p[i][v].z * q[i][v].z;
            part += p[i][v].w * q[i][v].w;
        }
        const float dot = row_sum16(part);

        float pred = gm;
        if (M::has_bu) pred += bu[i];
        if (M::has_bj) pred += bj[i];
        pred += dot;
        if (Traits<MODEL>::has_ctx) {
            float term = 0.f; // lane d carries the deviation of the tuple's d-th condition
            if (M::has_ic && M::has_uc) term = bic[i] + buc[i];
            else if (M::has_ic) term = bic[i];
            else if (M::has_uc) term = buc[i];
            pred += row_sum16(term);
        }
        const float e = rr[i] - pred;

        // scalar biases: lane 0 of the group owns the store
        if (l16 == 0) {
            if (COH) {
                if (M::has_bu) st_f32_coherent(a.userBias + uu[i], bu[i] + lr * (e - regB * bu[i]));
                if (M::has_bj) st_f32_coherent(a.itemBias + jj[i], bj[i] + lr * (e - regB * bj[i]));
            } else {
                if (M::has_bu) a.userBias[uu[i]] = bu[i] + lr * (e - regB * bu[i]);
                if (M::has_bj) a.itemBias[jj[i]] = bj[i] + lr * (e - regB * bj[i]);
            }
        }
        float ctx_loss = 0.f;
        if (cond[i] >= 0) {
            if (M::has_ic) {
                if (COH) st_f32_coherent(pic[i], bic[i] + lr * (e - regC * bic[i]));
                else *pic[i] = bic[i] + lr * (e - regC * bic[i]);
                ctx_loss += bic[i] * bic[i];
            }
            if (M::has_uc) {
                if (COH) st_f32_coherent(puc[i], buc[i] + lr * (e - regC * buc[i]));
                else *puc[i] = buc[i] + lr * (e - regC * buc[i]);
                ctx_loss += buc[i] * buc[i];
            }
        }

        float lsum = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            float4 pn, qn;
#define CMI_UPD(c)                                                                                       \
    pn.c = p[i][v].c + lr * (e * q[i][v].c - regU * p[i][v].c);                                          \
    qn.c = q[i][v].c + lr * (e * p[i][v].c - regI * q[i][v].c);                                          \
    lsum += (regU * p[i][v].c) * p[i][v].c + (regI * q[i][v].c) * q[i][v].c;
            CMI_UPD(x) CMI_UPD(y) CMI_UPD(z) CMI_UPD(w)
#undef CMI_UPD
            if (!RAGGED || 4 * l16 + 64 * v < K) {
                if (COH) {
                    f32x4 tp, tq;
                    tp.x = pn.x, tp.y = pn.y, tp.z = pn.z, tp.w = pn.w;
                    tq.x = qn.x, tq.y = qn.y, tq.z = qn.z, tq.w = qn.w;
                    st_row_coherent(rsP, (uint32_t)(((size_t)uu[i] * K + 4 * l16 + 64 * v) * 4), tp);
                    st_row_coherent(rsQ, (uint32_t)(((size_t)jj[i] * K + 4 * l16 + 64 * v) * 4), tq);
                } else {
                    prow[i][v * 16] = pn;
                    qrow[i][v * 16] = qn;
                }
            }
        }

        const float reg_loss = row_sum16(lsum);
        const float ctx_sum = (Traits<MODEL>::has_ctx) ? row_sum16(ctx_loss) : 0.f;
        if (l16 == 0) {
            double l = (double)e * (double)e;
            if (M::has_bu) l += (double)regB * bu[i] * bu[i];
            if (M::has_bj) l += (double)regB * bj[i] * bj[i];
            if (Traits<MODEL>::has_ctx) l += (double)regC * ctx_sum;
            gloss += l + (double)reg_loss;
        }
    }

    return gloss;
}

template <int MODEL, int VPL, int TPG, bool RAGGED = false, bool COH = false>
__global__ __launch_bounds__(256) void sgd_level_fast_f32(SgdArgs<float> a, int64_t begin, int count,
                                                          int64_t slot0) {
    __shared__ double s_loss[16];
    const int tid = threadIdx.x;
    const int l16 = tid & 15;
    const int gib = tid >> 4;
    // tuple i of this group: g0 + 16*i (keeps the stream loads coalesced)
    const double gloss = fast_tuples_f32<MODEL, VPL, TPG, RAGGED, COH, 16>(a, begin, count, blockIdx.x * (16 * TPG) + gib, l16);
    if (l16 == 0) s_loss[gib] = gloss;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += s_loss[i];
        a.loss_part[slot0 + blockIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// small-k fast path (fp32 state, any k <= 64 that is not 64: the reference's default num.factors is 10)
// ---------------------------------------------------------------------------------------------
// LPT lanes per tuple (4, 8 or 16 -> 16, 8 or 4 tuples per wave64), lane l owns factors l, l+LPT, l+2*LPT, l+3*LPT
// (scalar loads: rows of k floats are not 16-byte aligned for general k) and the tuple's l-th condition, so
// LPT >= dmax and 4*LPT >= k.  Same arithmetic as sgd_level_fast_f32; the group sums run on DPP quad permutes and
// row (half-)mirrors.  With the generic kernel a k=10 epoch of the C3 shape cost 1.8x the k=128 one (a whole wave per
// tuple, 10 of 64 lanes busy); this one moves the same few bytes with 16 tuples per wave.

template <int MODEL, int LPT, int TPG, int GS, bool PRE = false>
__device__ __forceinline__ double small_tuples_f32(const SgdArgs<float> &a, int64_t begin, int count, int g0, int lt,
                                                   const TupleIds *pre = nullptr) {
    using M = Traits<MODEL>;
    static_assert(MODEL != CAMF_C, "CAMF_C has no level schedule (shared condBias)");
    constexpr int VPL = 4;
    const int k = a.k;
    double gloss = 0.0;
    const HParams hp = *a.hp;
    const float lr = (float)hp.lr, regU = (float)hp.regU, regI = (float)hp.regI, regB = (float)hp.regB,
                regC = (float)hp.regC, gm = (float)hp.gm;

    bool live[TPG];
    int uu[TPG], jj[TPG], cond[TPG];
    float rr[TPG];
#pragma unroll
    for (int i = 0; i < TPG; ++i) {
        const TupleIds t = PRE ? *pre : load_tuple_ids<Traits<MODEL>::has_ctx>(a, begin, count, g0 + GS * i, lt);
        live[i] = t.live;
        uu[i] = t.uu;
        jj[i] = t.jj;
        rr[i] = t.rr;
        cond[i] = t.cond;
    }
    float *prow[TPG], *qrow[TPG];
    float p[TPG][VPL], q[TPG][VPL];
    float bu[TPG], bj[TPG], bic[TPG], buc[TPG];
    float *pic[TPG], *puc[TPG];
#pragma unroll
    for (int i = 0; i < TPG; ++i) {
        prow[i] = a.P + (size_t)uu[i] * k + lt;
        qrow[i] = a.Q + (size_t)jj[i] * k + lt;
        bu[i] = bj[i] = bic[i] = buc[i] = 0.f;
        pic[i] = puc[i] = nullptr;
#pragma unroll
        for (int v = 0; v < VPL; ++v) p[i][v] = q[i][v] = 0.f;
        if (live[i]) {
#pragma unroll
            for (int v = 0; v < VPL; ++v)
                if (lt + v * LPT < k) {
                    p[i][v] = prow[i][v * LPT];
                    q[i][v] = qrow[i][v * LPT];
                }
            if (M::has_bu) bu[i] = a.userBias[uu[i]];
            if (M::has_bj) bj[i] = a.itemBias[jj[i]];
            if (cond[i] >= 0) {
                if (M::has_ic) {
                    pic[i] = a.icBias + (size_t)jj[i] * a.n_conds + cond[i];
                    bic[i] = *pic[i];
                }
                if (M::has_uc) {
                    puc[i] = a.ucBias + (size_t)uu[i] * a.n_conds + cond[i];
                    buc[i] = *puc[i];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TPG; ++i) {
        if (!live[i]) continue; // group-uniform
        float part = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) part += p[i][v] * q[i][v];
        const float dot = group_sum<LPT>(part);
        float pred = gm;
        if (M::has_bu) pred += bu[i];
        if (M::has_bj) pred += bj[i];
        pred += dot;
        if (Traits<MODEL>::has_ctx) {
            float term = 0.f; // lane d carries the deviation of the tuple's d-th condition
            if (M::has_ic && M::has_uc) term = bic[i] + buc[i];
            else if (M::has_ic) term = bic[i];
            else if (M::has_uc) term = buc[i];
            pred += group_sum<LPT>(term);
        }
        const float e = rr[i] - pred;
        if (lt == 0) {
            if (M::has_bu) a.userBias[uu[i]] = bu[i] + lr * (e - regB * bu[i]);
            if (M::has_bj) a.itemBias[jj[i]] = bj[i] + lr * (e - regB * bj[i]);
        }
        float ctx_loss = 0.f;
        if (cond[i] >= 0) {
            if (M::has_ic) {
                *pic[i] = bic[i] + lr * (e - regC * bic[i]);
                ctx_loss += bic[i] * bic[i];
            }
            if (M::has_uc) {
                *puc[i] = buc[i] + lr * (e - regC * buc[i]);
                ctx_loss += buc[i] * buc[i];
            }
        }
        float lsum = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v)
            if (lt + v * LPT < k) {
                const float pv = p[i][v], qv = q[i][v];
                prow[i][v * LPT] = pv + lr * (e * qv - regU * pv);
                qrow[i][v * LPT] = qv + lr * (e * pv - regI * qv);
                lsum += (regU * pv) * pv + (regI * qv) * qv;
            }
        const float reg_loss = group_sum<LPT>(lsum);
        const float ctx_sum = (Traits<MODEL>::has_ctx) ? group_sum<LPT>(ctx_loss) : 0.f;
        if (lt == 0) {
            double l = (double)e * (double)e;
            if (M::has_bu) l += (double)regB * bu[i] * bu[i];
            if (M::has_bj) l += (double)regB * bj[i] * bj[i];
            if (Traits<MODEL>::has_ctx) l += (double)regC * ctx_sum;
            gloss += l + (double)reg_loss;
        }
    }
    return gloss;
}

template <int MODEL, int LPT, int TPG>
__global__ __launch_bounds__(256) void sgd_level_small_f32(SgdArgs<float> a, int64_t begin, int count, int64_t slot0) {
    constexpr int GPB = 256 / LPT; // groups per workgroup
    __shared__ double s_loss[GPB];
    const int tid = threadIdx.x;
    const int lt = tid % LPT;
    const int gib = tid / LPT;
    // tuple i of this group: g0 + GPB*i
    const double gloss = small_tuples_f32<MODEL, LPT, TPG, GPB>(a, begin, count, blockIdx.x * (GPB * TPG) + gib, lt);
    if (lt == 0) s_loss[gib] = gloss;
    __syncthreads();
    if (tid < 64) { // fixed-shape tree over the GPB group sums
        double s = 0.0;
        for (int g = tid; g < GPB; g += 64) s += s_loss[g];
        s = wave_sum64(s);
        if (tid == 0) a.loss_part[slot0 + blockIdx.x] = s;
    }
}


// ---------------------------------------------------------------------------------------------
// generic path: one wave64 per tuple, any k, T in {float,double}, optional strict order
// ---------------------------------------------------------------------------------------------

// Applies one update.  `loss` is the running sum the contributions are added to, in the
// reference's order when STRICT (loss += e*e; bias terms in source order; then one add per factor).
template <typename T, int MODEL, bool STRICT>
__device__ __forceinline__ double sgd_one(const SgdArgs<T> &a, const HParams &hp, int uu, int jj, T rr,
                                          const int32_t *conds, int lane, double loss) {
    using M = Traits<MODEL>;
    const int k = a.k;
    T *pu = a.P + (size_t)uu * k;
    T *qj = a.Q + (size_t)jj * k;
    const T lr = (T)hp.lr, regU = (T)hp.regU, regI = (T)hp.regI, regB = (T)hp.regB, regC = (T)hp.regC,
            gm = (T)hp.gm;

    // DenseMatrix.rowMult: s = 0; s += m[f]*n[f], f ascending (librec jar)
    T dot = 0;
    if (STRICT) {
        for (int c0 = 0; c0 < k; c0 += 64) {
            const int f = c0 + lane;
            const T prod = f < k ? pu[f] * qj[f] : (T)0;
            const int m = (k - c0) < 64 ? (k - c0) : 64;
            for (int l = 0; l < m; ++l) dot += __shfl(prod, l, 64);
        }
    } else {
        T part = 0;
        for (int f = lane; f < k; f += 64) part += pu[f] * qj[f];
        dot = wave_sum64(part);
    }

    // predict(): every lane evaluates the same scalar expression in the reference's add order
    T bu = 0, bj = 0;
    if (M::has_bu) bu = a.userBias[uu];
    if (M::has_bj) bj = a.itemBias[jj];
    T pred = gm;
    if (M::has_bu) pred += bu;
    if (M::has_bj) pred += bj;
    pred += dot;
    if (Traits<MODEL>::has_ctx) {
        for (int d = 0; d < a.dmax; ++d) {
            const int cond = conds[d];
            if (cond < 0) break;
            if (M::has_bc) pred += a.condBias[cond];
            if (M::has_ic && M::has_uc)
                pred += a.icBias[(size_t)jj * a.n_conds + cond] + a.ucBias[(size_t)uu * a.n_conds + cond];
            else if (M::has_ic)
                pred += a.icBias[(size_t)jj * a.n_conds + cond];
            else if (M::has_uc)
                pred += a.ucBias[(size_t)uu * a.n_conds + cond];
        }
    }
    const T e = rr - pred;
    loss += (double)(e * e);

    if (M::has_bu) {
        if (lane == 0) a.userBias[uu] = bu + lr * (e - regB * bu);
        loss += (double)((regB * bu) * bu);
    }
    if (M::has_bj) {
        if (lane == 0) a.itemBias[jj] = bj + lr * (e - regB * bj);
        loss += (double)((regB * bj) * bj);
    }
    if (Traits<MODEL>::has_ctx) {
        T s_ic = 0, s_uc = 0, s_bc = 0;
        for (int d = 0; d < a.dmax; ++d) {
            const int cond = conds[d];
            if (cond < 0) break;
            if (M::has_bc) {
                T *cell = a.condBias + cond;
                const T bc = *cell;
                s_bc += bc;
                if (lane == 0) *cell = bc + lr * (e - regC * bc);
            }
            if (M::has_uc) {
                T *cell = a.ucBias + (size_t)uu * a.n_conds + cond;
                const T b = *cell;
                s_uc += b * b;
                if (lane == 0) *cell = b + lr * (e - regC * b);
            }
            if (M::has_ic) {
                T *cell = a.icBias + (size_t)jj * a.n_conds + cond;
                const T b = *cell;
                s_ic += b * b;
                if (lane == 0) *cell = b + lr * (e - regC * b);
            }
        }
        if (M::has_bc) loss += (double)(regB * s_bc);                   // CAMF_C.java:115
        else if (M::has_ic && M::has_uc) loss += (double)(regC * s_ic + regC * s_uc); // CAMF_CUCI.java:111
        else if (M::has_ic) loss += (double)(regC * s_ic);
        else loss += (double)(regC * s_uc);
    }

    // factor loop: puf, qjf both read before either is written
    if (STRICT) {
        for (int c0 = 0; c0 < k; c0 += 64) {
            const int f = c0 + lane;
            T term = 0;
            if (f < k) {
                const T p = pu[f], q = qj[f];
                pu[f] = p + lr * (e * q - regU * p);
                qj[f] = q + lr * (e * p - regI * q);
                term = (regU * p) * p + (regI * q) * q;
            }
            const int m = (k - c0) < 64 ? (k - c0) : 64;
            for (int l = 0; l < m; ++l) loss += (double)__shfl(term, l, 64);
        }
    } else {
        T part = 0;
        for (int f = lane; f < k; f += 64) {
            const T p = pu[f], q = qj[f];
            pu[f] = p + lr * (e * q - regU * p);
            qj[f] = q + lr * (e * p - regI * q);
            part += (regU * p) * p + (regI * q) * q;
        }
        loss += wave_sum64((double)part);
    }
    return loss;
}

template <typename T, int MODEL, bool STRICT>
__global__ __launch_bounds__(256) void sgd_level_generic(SgdArgs<T> a, int64_t begin, int count, int64_t slot0) {
    __shared__ double s_loss[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + wave;
    double gl = 0.0;
    if (g < count) {
        const int64_t t = begin + g;
        const HParams hp = *a.hp;
        gl = sgd_one<T, MODEL, STRICT>(a, hp, a.su[t], a.sj[t], a.sr[t], a.sconds + t * a.dmax, lane, 0.0);
    }
    if (lane == 0) s_loss[wave] = gl;
    __syncthreads();
    if (threadIdx.x == 0) a.loss_part[slot0 + blockIdx.x] = ((s_loss[0] + s_loss[1]) + s_loss[2]) + s_loss[3];
}

// The narrow tail of a schedule: with heavy-tailed item (or user) degrees the longest chains belong to a few hot rows
// and the last levels hold a handful of tuples each -- hundreds of thousands of them for a Zipf item distribution.
// One launch per such level would cost ~7 us apiece; here ONE workgroup walks all tail levels inside one launch:
// wave w takes tuples w, w+16, ... of the level, a workgroup barrier separates levels.  All 16 waves share the CU's
// vector L1 and its XCD's L2, so the barrier's workgroup-scope ordering is all the coherence that is needed, and the
// hot rows stay cache-resident from one level to the next.
template <typename T, int MODEL, bool STRICT>
__global__ __launch_bounds__(1024) void sgd_tail_kernel(SgdArgs<T> a, const int64_t *__restrict__ tail_off, int n_tail,
                                                        int64_t slot) {
    __shared__ double s_loss[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const HParams hp = *a.hp;
    double gl = 0.0;
    int64_t b = tail_off[0];
    for (int l = 0; l < n_tail; ++l) {
        const int64_t e = tail_off[l + 1];
        for (int64_t t = b + wave; t < e; t += 16)
            gl += sgd_one<T, MODEL, STRICT>(a, hp, a.su[t], a.sj[t], a.sr[t], a.sconds + t * a.dmax, lane, 0.0);
        b = e;
        __syncthreads(); // release/acquire at workgroup scope: the next level sees this level's rows
    }
    if (lane == 0) s_loss[wave] = gl;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) sum += s_loss[w];
        a.loss_part[slot] = sum;
    }
}

// the same walk with the arithmetic (and hence the bits) of the level kernels it stands in for: 64 sixteen-lane groups
template <int MODEL, int VPL, bool RAGGED>
__global__ __launch_bounds__(1024) void sgd_tail_fast_f32(SgdArgs<float> a, const int64_t *__restrict__ tail_off, int n_tail,
                                                          int64_t slot) {
    __shared__ double s_loss[64];
    const int l16 = threadIdx.x & 15, gib = threadIdx.x >> 4;
    double gl = 0.0;
    int64_t b = tail_off[0];
    // the tuple ids of a level do not depend on the previous level's updates: they are fetched one level ahead, which
    // takes one of the two dependent memory round trips (ids -> rows) off every level's critical path
    int64_t e = tail_off[1];
    TupleIds next = load_tuple_ids<Traits<MODEL>::has_ctx>(a, b, (int)(e - b), gib, l16);
    for (int l = 0; l < n_tail; ++l) {
        const int cnt = (int)(e - b);
        const TupleIds cur = next;
        const int64_t e2 = l + 1 < n_tail ? tail_off[l + 2] : e;
        if (l + 1 < n_tail) next = load_tuple_ids<Traits<MODEL>::has_ctx>(a, e, (int)(e2 - e), gib, l16);
        gl += fast_tuples_f32<MODEL, VPL, 1, RAGGED, false, 64, true>(a, b, cnt, gib, l16, &cur);
        for (int base = 64; base < cnt; base += 64) gl += fast_tuples_f32<MODEL, VPL, 1, RAGGED, false, 64>(a, b, cnt, base + gib, l16);
        b = e;
        e = e2;
        __syncthreads();
    }
    if (l16 == 0) s_loss[gib] = gl;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
        for (int g = 0; g < 64; ++g) sum += s_loss[g];
        a.loss_part[slot] = sum;
    }
}

template <int MODEL, int LPT>
__global__ __launch_bounds__(1024) void sgd_tail_small_f32(SgdArgs<float> a, const int64_t *__restrict__ tail_off, int n_tail,
                                                           int64_t slot) {
    constexpr int G = 1024 / LPT;
    __shared__ double s_loss[G];
    const int lt = threadIdx.x % LPT, gib = threadIdx.x / LPT;
    double gl = 0.0;
    int64_t b = tail_off[0];
    int64_t e = tail_off[1];
    TupleIds next = load_tuple_ids<Traits<MODEL>::has_ctx>(a, b, (int)(e - b), gib, lt);
    for (int l = 0; l < n_tail; ++l) { // ids one level ahead, as in sgd_tail_fast_f32
        const int cnt = (int)(e - b);
        const TupleIds cur = next;
        const int64_t e2 = l + 1 < n_tail ? tail_off[l + 2] : e;
        if (l + 1 < n_tail) next = load_tuple_ids<Traits<MODEL>::has_ctx>(a, e, (int)(e2 - e), gib, lt);
        gl += small_tuples_f32<MODEL, LPT, 1, G, true>(a, b, cnt, gib, lt, &cur);
        for (int base = G; base < cnt; base += G) gl += small_tuples_f32<MODEL, LPT, 1, G>(a, b, cnt, base + gib, lt);
        b = e;
        e = e2;
        __syncthreads();
    }
    if (lt == 0) s_loss[gib] = gl;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
        for (int g = 0; g < G; ++g) sum += s_loss[g];
        a.loss_part[slot] = sum;
    }
}

// One wavefront, tuples strictly in stream order: the reference's single-threaded semantics.
// Same-wave stores and later loads of the same address stay ordered in the vector memory pipeline.
template <typename T, int MODEL, bool STRICT>
__global__ __launch_bounds__(64) void sgd_serial(SgdArgs<T> a, int64_t n, double *loss_out) {
    const int lane = threadIdx.x;
    const HParams hp = *a.hp;
    double loss = 0.0;
    for (int64_t base = 0; base < n; base += 64) {
        const int64_t t = base + lane;
        int mu = 0, mj = 0;
        T mr = 0;
        if (t < n) {
            mu = a.su[t];
            mj = a.sj[t];
            mr = a.sr[t];
        }
        const int m = (n - base) < 64 ? (int)(n - base) : 64;
        for (int i = 0; i < m; ++i) {
            const int uu = __shfl(mu, i, 64), jj = __shfl(mj, i, 64);
            const T rr = __shfl(mr, i, 64);
            loss = sgd_one<T, MODEL, STRICT>(a, hp, uu, jj, rr, a.sconds + (base + i) * a.dmax, lane, loss);
        }
    }
    if (lane == 0) loss_out[0] = loss * 0.5;
}

// ---------------------------------------------------------------------------------------------
// fast serial path: one wave64, rows in registers, next tuple prefetched, condBias in LDS
// ---------------------------------------------------------------------------------------------
//
// The reference order is one dependency chain, so the only lever is latency per tuple.  Compared with
// sgd_serial (two passes over the rows, a chain of dependent global loads per condition) this kernel
//   * keeps the k <= 256 factors of P[u], Q[j] in registers (lane f owns f, f+64, f+128, f+192);
//   * issues the loads of tuple t+1 (rows, scalar biases) BEFORE computing tuple t; if t+1 shares the user or
//     the item with t the stale prefetch is dropped and the freshly updated registers are forwarded instead;
//   * stages 64 tuples and their condition ids per chunk (one coalesced load each) and keeps CAMF_C's whole
//     condBias vector in LDS for the kernel's lifetime (it is what makes CAMF_C sequential);
//   * reduces the dot with DPP (row_ror 8/4/2/1, row_bcast15, row_bcast31) instead of six bpermutes.
// Arithmetic per element is the same expression as everywhere else; the dot and the loss are tree sums
// (not strict).  Same-wave store -> later load of one address stays ordered in the vector memory pipeline.

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_rows_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_rows_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// v_readlane with a wave-uniform lane index: the value lands in an SGPR (a few cycles) instead of going through
// the LDS crossbar like ds_bpermute (__shfl)
__device__ __forceinline__ int rl(int x, int lane) { return __builtin_amdgcn_readlane(x, lane); }
__device__ __forceinline__ float rl(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
__device__ __forceinline__ double rl(double x, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}

// wave64 sum, result uniform (read from lane 63); fixed tree
__device__ __forceinline__ float wave_sum_dpp(float x) {
    x += dpp_rows_f32<0x128, 0xf>(x);
    x += dpp_rows_f32<0x124, 0xf>(x);
    x += dpp_rows_f32<0x122, 0xf>(x);
    x += dpp_rows_f32<0x121, 0xf>(x);       // every lane: its row's sum
    x += dpp_rows_f32<0x142, 0xa>(x);       // row_bcast15 into rows 1,3
    x += dpp_rows_f32<0x143, 0xc>(x);       // row_bcast31 into rows 2,3
    return rl(x, 63);
}
__device__ __forceinline__ double wave_sum_dpp(double x) {
    x += dpp_rows_f64<0x128, 0xf>(x);
    x += dpp_rows_f64<0x124, 0xf>(x);
    x += dpp_rows_f64<0x122, 0xf>(x);
    x += dpp_rows_f64<0x121, 0xf>(x);
    x += dpp_rows_f64<0x142, 0xa>(x);
    x += dpp_rows_f64<0x143, 0xc>(x);
    return rl(x, 63);
}

// FULL: k == 64*MAXC, so no factor lane is ever masked -- together with the unconditional prefetch and the
// all-lane bias stores this leaves the inner loop free of memory operations under a branch, and the compiler can
// count outstanding loads/stores exactly (s_waitcnt vmcnt(N) instead of vmcnt(0)): the wait for the prefetched
// rows no longer drains this tuple's stores.
template <typename T, int MODEL, int MAXC, bool FULL> // k <= 64 * MAXC
__global__ __launch_bounds__(64) void sgd_serial_fast(SgdArgs<T> a, int64_t n, double *loss_out) {
    using M = Traits<MODEL>;
    extern __shared__ unsigned char smem_raw[];
    T *s_bc = reinterpret_cast<T *>(smem_raw);                                   // [n_conds] (CAMF_C)
    int32_t *s_conds = reinterpret_cast<int32_t *>(s_bc + (M::has_bc ? a.n_conds : 0)); // [64 x dmax] chunk
    const int lane = threadIdx.x;
    const int k = a.k, dmax = a.dmax;
    const HParams hp = *a.hp;
    const T lr = (T)hp.lr, regU = (T)hp.regU, regI = (T)hp.regI, regB = (T)hp.regB, regC = (T)hp.regC, gm = (T)hp.gm;
    if (M::has_bc)
        for (int c = lane; c < a.n_conds; c += 64) s_bc[c] = a.condBias[c];
    __syncthreads();

    double loss = 0.0;
    T p[MAXC], q[MAXC], pn[MAXC], qn[MAXC]; // current rows / prefetched rows of the next tuple
    T bu = 0, bj = 0, bu_n = 0, bj_n = 0;
    int cu = -1, cj = -1; // user / item whose rows are in p, q
    if (n > 0) { // rows of the very first tuple
        const int u0 = a.su[0], j0 = a.sj[0];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int f = lane + 64 * c;
            pn[c] = (FULL || f < k) ? a.P[(size_t)u0 * k + (FULL ? f : (f < k ? f : 0))] : (T)0;
            qn[c] = (FULL || f < k) ? a.Q[(size_t)j0 * k + (FULL ? f : (f < k ? f : 0))] : (T)0;
        }
        if (M::has_bu) bu_n = a.userBias[u0];
        if (M::has_bj) bj_n = a.itemBias[j0];
    }

    for (int64_t base = 0; base < n; base += 64) {
        const int m = (n - base) < 64 ? (int)(n - base) : 64;
        int mu = 0, mj = 0;
        T mr = 0;
        if (lane < m) {
            mu = a.su[base + lane];
            mj = a.sj[base + lane];
            mr = a.sr[base + lane];
        }
        int next_u0 = 0, next_j0 = 0; // first tuple of the next chunk (for the prefetch of this chunk's last tuple)
        if (base + 64 < n) {
            next_u0 = a.su[base + 64];
            next_j0 = a.sj[base + 64];
        }
        if (Traits<MODEL>::has_ctx) {
            __syncthreads();
            for (int x = lane; x < m * dmax; x += 64) s_conds[x] = a.sconds[base * dmax + x];
            __syncthreads();
        }
        for (int i = 0; i < m; ++i) {
            const int uu = rl(mu, i), jj = rl(mj, i);
            const T rr = rl(mr, i);
            // ---- rows of this tuple: forwarded registers, or the prefetch, or (first tuple) a fresh load
            if (uu != cu) {
#pragma unroll
                for (int c = 0; c < MAXC; ++c) p[c] = pn[c];
                bu = bu_n;
            }
            if (jj != cj) {
#pragma unroll
                for (int c = 0; c < MAXC; ++c) q[c] = qn[c];
                bj = bj_n;
            }
            cu = uu;
            cj = jj;
            // ---- prefetch the next tuple (the last tuple of a chunk peeks into the next chunk)
            // (unconditional: the last tuple of all re-reads its own rows, which are then simply not used)
            int nu = uu, nj = jj;
            if (i + 1 < m) {
                nu = rl(mu, i + 1);
                nj = rl(mj, i + 1);
            } else if (base + 64 < n) {
                nu = next_u0;
                nj = next_j0;
            }
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int f = lane + 64 * c;
                if (FULL) {
                    pn[c] = a.P[(size_t)nu * k + f];
                    qn[c] = a.Q[(size_t)nj * k + f];
                } else {
                    pn[c] = f < k ? a.P[(size_t)nu * k + f] : (T)0;
                    qn[c] = f < k ? a.Q[(size_t)nj * k + f] : (T)0;
                }
            }
            if (M::has_bu) bu_n = a.userBias[nu];
            if (M::has_bj) bj_n = a.itemBias[nj];
            // ---- condition of this lane (lane d < dmax) and its bias entries
            int cond = -1;
            T bc = 0, bic = 0, buc = 0;
            T *pic = nullptr, *puc = nullptr;
            if (Traits<MODEL>::has_ctx && lane < dmax) cond = s_conds[i * dmax + lane];
            if (cond >= 0) {
                if (M::has_bc) bc = s_bc[cond];
                if (M::has_ic) {
                    pic = a.icBias + (size_t)jj * a.n_conds + cond;
                    bic = *pic;
                }
                if (M::has_uc) {
                    puc = a.ucBias + (size_t)uu * a.n_conds + cond;
                    buc = *puc;
                }
            }
            // ---- predict
            T part = 0;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) part += p[c] * q[c];
            const T dot = wave_sum_dpp(part);
            T pred = gm;
            if (M::has_bu) pred += bu;
            if (M::has_bj) pred += bj;
            pred += dot;
            if (Traits<MODEL>::has_ctx) {
                T term = 0;
                if (M::has_bc) term = bc;
                else if (M::has_ic && M::has_uc) term = bic + buc;
                else if (M::has_ic) term = bic;
                else term = buc;
                const unsigned long long present = __ballot(cond >= 0);
                for (int d = 0; d < dmax; ++d) // the reference adds the deviations one by one, in condition order
                    if ((present >> d) & 1ull) pred += rl(term, d);
            }
            const T e = rr - pred;
            // ---- biases
            double l = (double)(e * e);
            if (M::has_bu) {
                const T nb = bu + lr * (e - regB * bu);
                a.userBias[uu] = nb; // every lane stores the same value to the same word: no branch in the loop
                l += (double)((regB * bu) * bu);
                bu = nb;
            }
            if (M::has_bj) {
                const T nb = bj + lr * (e - regB * bj);
                a.itemBias[jj] = nb;
                l += (double)((regB * bj) * bj);
                bj = nb;
            }
            T ctx_term = 0;
            if (cond >= 0) {
                if (M::has_bc) {
                    s_bc[cond] = bc + lr * (e - regC * bc);
                    ctx_term = bc; // plain sum, weighted by regB (reference quirk, CAMF_C.java:110,115)
                }
                if (M::has_ic) {
                    *pic = bic + lr * (e - regC * bic);
                    ctx_term += bic * bic;
                }
                if (M::has_uc) {
                    *puc = buc + lr * (e - regC * buc);
                    ctx_term += buc * buc;
                }
            }
            // ---- factors
            T reg_part = 0;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int f = lane + 64 * c;
                const T pv = p[c], qv = q[c];
                p[c] = pv + lr * (e * qv - regU * pv);
                q[c] = qv + lr * (e * pv - regI * qv);
                reg_part += (regU * pv) * pv + (regI * qv) * qv;
                if (FULL || f < k) {
                    a.P[(size_t)uu * k + f] = p[c];
                    a.Q[(size_t)jj * k + f] = q[c];
                }
            }
            if (Traits<MODEL>::has_ctx) l += (double)((M::has_bc ? regB : regC) * wave_sum_dpp(ctx_term));
            l += (double)wave_sum_dpp(reg_part);
            loss += l;
        }
    }
    if (M::has_bc) {
        __syncthreads();
        for (int c = lane; c < a.n_conds; c += 64) a.condBias[c] = s_bc[c];
    }
    if (lane == 0) loss_out[0] = loss * 0.5;
}

// ---------------------------------------------------------------------------------------------
// small utility kernels
// ---------------------------------------------------------------------------------------------

__global__ void set_hparams_kernel(HParams *dst, HParams v) { *dst = v; }

// Fixed-shape two-stage reduction: the result depends only on (n_slots), never on timing.
__global__ __launch_bounds__(256) void reduce_loss_stage1(const double *part, int64_t n, double *scratch) {
    __shared__ double s[256];
    const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t b = (int64_t)blockIdx.x * chunk;
    const int64_t e = (b + chunk) < n ? (b + chunk) : n;
    double acc = 0.0;
    for (int64_t i = b + threadIdx.x; i < e; i += 256) acc += part[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) scratch[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(256) void reduce_loss_stage2(const double *scratch, int nblk, double *loss_out) {
    __shared__ double s[256];
    s[threadIdx.x] = (int)threadIdx.x < nblk ? scratch[threadIdx.x] : 0.0;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[0] = s[0] * 0.5;
}

template <typename S, typename D>
__global__ void convert_kernel(const S *src, D *dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (D)src[i];
}

// ---------------------------------------------------------------------------------------------
// multi-GPU exchange of the replicated item-side state (carskit_amd/dist.py): two elementwise passes around the collective
// ---------------------------------------------------------------------------------------------
// pack:  bucket = state - snapshot                       (this rank's movement during the epoch)
// apply: state = snapshot + scale * bucket; snapshot = state   (bucket now holds the sum over ranks)
template <typename T>
__global__ __launch_bounds__(256) void delta_pack_kernel(const T *__restrict__ state, const T *__restrict__ snap, T *__restrict__ bucket,
                                                         int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bucket[i] = state[i] - snap[i];
}
template <typename T>
__global__ __launch_bounds__(256) void delta_apply_kernel(T *__restrict__ state, T *__restrict__ snap, const T *__restrict__ bucket, T scale,
                                                          int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T v = snap[i] + scale * bucket[i];
        state[i] = v;
        snap[i] = v;
    }
}
// the float4 forms move 16 B per lane (segments are 16-byte aligned and padded to 4 elements on the host side)
__global__ __launch_bounds__(256) void delta_pack_f32x4(const float4 *__restrict__ state, const float4 *__restrict__ snap,
                                                        float4 *__restrict__ bucket, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = state[i], b = snap[i];
        bucket[i] = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    }
}
__global__ __launch_bounds__(256) void delta_apply_f32x4(float4 *__restrict__ state, float4 *__restrict__ snap, const float4 *__restrict__ bucket,
                                                         float scale, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 b = snap[i], d = bucket[i];
        const float4 v = make_float4(b.x + scale * d.x, b.y + scale * d.y, b.z + scale * d.z, b.w + scale * d.w);
        state[i] = v;
        snap[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// predict / evalRatings (Recommender.java:306-317, 504-594): fp64 arithmetic over the stored state
// ---------------------------------------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(256) void eval_kernel(EvalArgs<T> a, int64_t n) {
    __shared__ double s_part[4][5];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * 4;
    double s_abs = 0, s_sq = 0, s_rabs = 0, s_rsq = 0, s_cnt = 0;
    const int k = a.k;
    const int model = a.model;
    const bool has_bu = model == BIASEDMF || model == CAMF_C || model == CAMF_CI;
    const bool has_bj = model == BIASEDMF || model == CAMF_C || model == CAMF_CU;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n; t += stride) {
        const int uu = a.u[t], jj = a.j[t];
        const T *pu = a.P + (size_t)uu * k;
        const T *qj = a.Q + (size_t)jj * k;
        double part = 0.0;
        for (int f = lane; f < k; f += 64) part += (double)pu[f] * (double)qj[f];
        const double dot = wave_sum64(part);
        double pred = a.gm;
        if (has_bu) pred += (double)a.userBias[uu];
        if (has_bj) pred += (double)a.itemBias[jj];
        pred += dot;
        if (model != BIASEDMF && model != PMF) {
            const int c = a.ctx[t];
            for (int q = a.ctx_ptr[c]; q < a.ctx_ptr[c + 1]; ++q) {
                const int cond = a.ctx_conds[q];
                if (model == CAMF_C) pred += (double)a.condBias[cond];
                else if (model == CAMF_CI) pred += (double)a.icBias[(size_t)jj * a.n_conds + cond];
                else if (model == CAMF_CU) pred += (double)a.ucBias[(size_t)uu * a.n_conds + cond];
                else pred += (double)a.icBias[(size_t)jj * a.n_conds + cond] + (double)a.ucBias[(size_t)uu * a.n_conds + cond];
            }
        }
        if (a.bound) {
            if (pred > a.hi) pred = a.hi;
            if (pred < a.lo) pred = a.lo;
        }
        if (a.preds && lane == 0) a.preds[t] = pred;
        if (a.r && !isnan(pred)) {
            const double rate = a.r[t];
            const double rpred = floor(pred / a.min_rate + 0.5) * a.min_rate; // Math.round(x)*minRate
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
// host-side launchers
// ---------------------------------------------------------------------------------------------

static bool level_coherent() { // experiment: device-coherent model traffic in the level kernel (k = 128, tables < 4 GiB)
    static const bool on = cmi_exp_env("CMI_LEVEL_COHERENT") != nullptr;
    return on;
}
static int g_fast_tpg = -1;
static int fast_tpg() { // tuples per 16-lane group (CMI_LEVEL_TPG overrides for experiments)
    if (g_fast_tpg < 0) {
        int v = 2;
        if (const char *env = cmi_exp_env("CMI_LEVEL_TPG")) v = atoi(env);
        g_fast_tpg = (v == 1 || v == 2 || v == 4) ? v : 2;
    }
    return g_fast_tpg;
}

int level_blocks_f32_fast(int, int count) { return (count + 16 * fast_tpg() - 1) / (16 * fast_tpg()); }
int level_blocks_generic(int count) { return (count + 3) / 4; }

bool has_fast_path(int k, int dmax, bool f64, const LaunchCfg &cfg) {
    if (f64 || cfg.strict) return false;
    if (k < 64 || k > 256 || k % 4 != 0) return false; // k = 64/128/256: exact kernels; other multiples of 4: masked float4 slots
    if (dmax > 16) return false;
    return true;
}

// small-k path: lanes per tuple
int small_lpt(int k, int dmax) {
    if (k <= 16 && dmax <= 4) return 4;
    if (k <= 32 && dmax <= 8) return 8;
    return 16;
}
bool has_small_path(int k, int dmax, bool f64, const LaunchCfg &cfg) {
    if (f64 || cfg.strict || cfg.model == CAMF_C) return false;
    if (cmi_exp_env("CMI_NO_SMALL_K")) return false; // A/B experiments
    return k < 64 && dmax <= 16;
}
constexpr int SMALL_TPG = 2;
int level_blocks_small(int k, int dmax, int count) {
    const int per = (256 / small_lpt(k, dmax)) * SMALL_TPG;
    return (count + per - 1) / per;
}
template <int MODEL>
static hipError_t launch_small_model(const SgdArgs<float> &a, int64_t begin, int count, int64_t slot0, hipStream_t s) {
    const dim3 grid(level_blocks_small(a.k, a.dmax, count)), block(256);
    switch (small_lpt(a.k, a.dmax)) {
    case 4: hipLaunchKernelGGL((sgd_level_small_f32<MODEL, 4, SMALL_TPG>), grid, block, 0, s, a, begin, count, slot0); break;
    case 8: hipLaunchKernelGGL((sgd_level_small_f32<MODEL, 8, SMALL_TPG>), grid, block, 0, s, a, begin, count, slot0); break;
    default: hipLaunchKernelGGL((sgd_level_small_f32<MODEL, 16, SMALL_TPG>), grid, block, 0, s, a, begin, count, slot0); break;
    }
    return hipGetLastError();
}
hipError_t launch_level_small_f32(const SgdArgs<float> &a, const LaunchCfg &cfg, int64_t begin, int count, int64_t slot0,
                                  hipStream_t s) {
    if (count <= 0) return hipSuccess;
    switch (cfg.model) {
    case BIASEDMF: return launch_small_model<BIASEDMF>(a, begin, count, slot0, s);
    case PMF: return launch_small_model<PMF>(a, begin, count, slot0, s);
    case CAMF_CI: return launch_small_model<CAMF_CI>(a, begin, count, slot0, s);
    case CAMF_CU: return launch_small_model<CAMF_CU>(a, begin, count, slot0, s);
    case CAMF_CUCI: return launch_small_model<CAMF_CUCI>(a, begin, count, slot0, s);
    }
    return hipErrorInvalidValue;
}

template <int MODEL, int TPG>
static hipError_t launch_fast_model_tpg(const SgdArgs<float> &a, int64_t begin, int count, int64_t slot0,
                                        hipStream_t s) {
    const dim3 grid((count + 16 * TPG - 1) / (16 * TPG)), block(256);
    switch (a.k) {
    case 64: hipLaunchKernelGGL((sgd_level_fast_f32<MODEL, 1, TPG>), grid, block, 0, s, a, begin, count, slot0); break;
    case 128:
        if (level_coherent()) hipLaunchKernelGGL((sgd_level_fast_f32<MODEL, 2, TPG, false, true>), grid, block, 0, s, a, begin, count, slot0);
        else hipLaunchKernelGGL((sgd_level_fast_f32<MODEL, 2, TPG>), grid, block, 0, s, a, begin, count, slot0);
        break;
    case 256: hipLaunchKernelGGL((sgd_level_fast_f32<MODEL, 4, TPG>), grid, block, 0, s, a, begin, count, slot0); break;
    default: // ragged k (multiple of 4)
        if (a.k < 128) hipLaunchKernelGGL((sgd_level_fast_f32<MODEL, 2, TPG, true>), grid, block, 0, s, a, begin, count, slot0);
        else if (a.k < 192) hipLaunchKernelGGL((sgd_level_fast_f32<MODEL, 3, TPG, true>), grid, block, 0, s, a, begin, count, slot0);
        else hipLaunchKernelGGL((sgd_level_fast_f32<MODEL, 4, TPG, true>), grid, block, 0, s, a, begin, count, slot0);
        break;
    }
    return hipGetLastError();
}

template <int MODEL>
static hipError_t launch_fast_model(const SgdArgs<float> &a, const LaunchCfg &, int64_t begin, int count,
                                    int64_t slot0, hipStream_t s) {
    switch (fast_tpg()) {
    case 1: return launch_fast_model_tpg<MODEL, 1>(a, begin, count, slot0, s);
    case 4: return launch_fast_model_tpg<MODEL, 4>(a, begin, count, slot0, s);
    default: return launch_fast_model_tpg<MODEL, 2>(a, begin, count, slot0, s);
    }
}

template <int MODEL, int TPG>
static void *fast_kernel_ptr(int k) {
    switch (k) {
    case 64: return (void *)sgd_level_fast_f32<MODEL, 1, TPG>;
    case 128: return level_coherent() ? (void *)sgd_level_fast_f32<MODEL, 2, TPG, false, true> : (void *)sgd_level_fast_f32<MODEL, 2, TPG>;
    case 256: return (void *)sgd_level_fast_f32<MODEL, 4, TPG>;
    }
    if (k > 64 && k < 256 && k % 4 == 0) {
        if (k < 128) return (void *)sgd_level_fast_f32<MODEL, 2, TPG, true>;
        if (k < 192) return (void *)sgd_level_fast_f32<MODEL, 3, TPG, true>;
        return (void *)sgd_level_fast_f32<MODEL, 4, TPG, true>;
    }
    return nullptr;
}
template <int MODEL>
static void *fast_kernel_ptr_model(int k) {
    switch (fast_tpg()) {
    case 1: return fast_kernel_ptr<MODEL, 1>(k);
    case 4: return fast_kernel_ptr<MODEL, 4>(k);
    default: return fast_kernel_ptr<MODEL, 2>(k);
    }
}

hipError_t launch_level_fast_f32(const SgdArgs<float> &a, const LaunchCfg &cfg, int64_t begin, int count,
                                 int64_t slot0, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    switch (cfg.model) {
    case BIASEDMF: return launch_fast_model<BIASEDMF>(a, cfg, begin, count, slot0, s);
    case PMF: return launch_fast_model<PMF>(a, cfg, begin, count, slot0, s);
    case CAMF_CI: return launch_fast_model<CAMF_CI>(a, cfg, begin, count, slot0, s);
    case CAMF_CU: return launch_fast_model<CAMF_CU>(a, cfg, begin, count, slot0, s);
    case CAMF_CUCI: return launch_fast_model<CAMF_CUCI>(a, cfg, begin, count, slot0, s);
    }
    return hipErrorInvalidValue;
}


template <typename T, int MODEL>
static hipError_t launch_generic_model(const SgdArgs<T> &a, const LaunchCfg &cfg, int64_t begin, int count,
                                       int64_t slot0, hipStream_t s) {
    const dim3 grid(level_blocks_generic(count)), block(256);
    if (cfg.strict)
        hipLaunchKernelGGL((sgd_level_generic<T, MODEL, true>), grid, block, 0, s, a, begin, count, slot0);
    else
        hipLaunchKernelGGL((sgd_level_generic<T, MODEL, false>), grid, block, 0, s, a, begin, count, slot0);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_tail(const SgdArgs<T> &a, const LaunchCfg &cfg, const int64_t *tail_off, int n_tail, int64_t slot,
                       hipStream_t s);
template <int MODEL>
static hipError_t launch_tail_fast_model(const SgdArgs<float> &a, const int64_t *tail_off, int n_tail, int64_t slot, hipStream_t s) {
    const dim3 g(1), b(1024);
    switch (a.k) {
    case 64: hipLaunchKernelGGL((sgd_tail_fast_f32<MODEL, 1, false>), g, b, 0, s, a, tail_off, n_tail, slot); break;
    case 128: hipLaunchKernelGGL((sgd_tail_fast_f32<MODEL, 2, false>), g, b, 0, s, a, tail_off, n_tail, slot); break;
    case 256: hipLaunchKernelGGL((sgd_tail_fast_f32<MODEL, 4, false>), g, b, 0, s, a, tail_off, n_tail, slot); break;
    default:
        if (a.k < 128) hipLaunchKernelGGL((sgd_tail_fast_f32<MODEL, 2, true>), g, b, 0, s, a, tail_off, n_tail, slot);
        else if (a.k < 192) hipLaunchKernelGGL((sgd_tail_fast_f32<MODEL, 3, true>), g, b, 0, s, a, tail_off, n_tail, slot);
        else hipLaunchKernelGGL((sgd_tail_fast_f32<MODEL, 4, true>), g, b, 0, s, a, tail_off, n_tail, slot);
    }
    return hipGetLastError();
}
template <int MODEL>
static hipError_t launch_tail_small_model(const SgdArgs<float> &a, const int64_t *tail_off, int n_tail, int64_t slot, hipStream_t s) {
    const dim3 g(1), b(1024);
    switch (small_lpt(a.k, a.dmax)) {
    case 4: hipLaunchKernelGGL((sgd_tail_small_f32<MODEL, 4>), g, b, 0, s, a, tail_off, n_tail, slot); break;
    case 8: hipLaunchKernelGGL((sgd_tail_small_f32<MODEL, 8>), g, b, 0, s, a, tail_off, n_tail, slot); break;
    default: hipLaunchKernelGGL((sgd_tail_small_f32<MODEL, 16>), g, b, 0, s, a, tail_off, n_tail, slot); break;
    }
    return hipGetLastError();
}
// kind: 0 generic (any k / fp64 / strict), 1 the float4 kernels' arithmetic, 2 the small-k kernels' arithmetic
hipError_t launch_tail_f32(const SgdArgs<float> &a, const LaunchCfg &cfg, int kind, const int64_t *tail_off, int n_tail,
                           int64_t slot, hipStream_t s) {
    if (n_tail <= 0) return hipSuccess;
    if (kind == 0) return launch_tail<float>(a, cfg, tail_off, n_tail, slot, s);
    switch (cfg.model) {
    case BIASEDMF: return kind == 1 ? launch_tail_fast_model<BIASEDMF>(a, tail_off, n_tail, slot, s) : launch_tail_small_model<BIASEDMF>(a, tail_off, n_tail, slot, s);
    case PMF: return kind == 1 ? launch_tail_fast_model<PMF>(a, tail_off, n_tail, slot, s) : launch_tail_small_model<PMF>(a, tail_off, n_tail, slot, s);
    case CAMF_CI: return kind == 1 ? launch_tail_fast_model<CAMF_CI>(a, tail_off, n_tail, slot, s) : launch_tail_small_model<CAMF_CI>(a, tail_off, n_tail, slot, s);
    case CAMF_CU: return kind == 1 ? launch_tail_fast_model<CAMF_CU>(a, tail_off, n_tail, slot, s) : launch_tail_small_model<CAMF_CU>(a, tail_off, n_tail, slot, s);
    case CAMF_CUCI: return kind == 1 ? launch_tail_fast_model<CAMF_CUCI>(a, tail_off, n_tail, slot, s) : launch_tail_small_model<CAMF_CUCI>(a, tail_off, n_tail, slot, s);
    }
    return hipErrorInvalidValue;
}

template <typename T, int MODEL>
static hipError_t launch_tail_model(const SgdArgs<T> &a, const LaunchCfg &cfg, const int64_t *tail_off, int n_tail,
                                    int64_t slot, hipStream_t s) {
    if (cfg.strict) hipLaunchKernelGGL((sgd_tail_kernel<T, MODEL, true>), dim3(1), dim3(1024), 0, s, a, tail_off, n_tail, slot);
    else hipLaunchKernelGGL((sgd_tail_kernel<T, MODEL, false>), dim3(1), dim3(1024), 0, s, a, tail_off, n_tail, slot);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_tail(const SgdArgs<T> &a, const LaunchCfg &cfg, const int64_t *tail_off, int n_tail, int64_t slot,
                       hipStream_t s) {
    if (n_tail <= 0) return hipSuccess;
    switch (cfg.model) {
    case BIASEDMF: return launch_tail_model<T, BIASEDMF>(a, cfg, tail_off, n_tail, slot, s);
    case PMF: return launch_tail_model<T, PMF>(a, cfg, tail_off, n_tail, slot, s);
    case CAMF_CI: return launch_tail_model<T, CAMF_CI>(a, cfg, tail_off, n_tail, slot, s);
    case CAMF_CU: return launch_tail_model<T, CAMF_CU>(a, cfg, tail_off, n_tail, slot, s);
    case CAMF_CUCI: return launch_tail_model<T, CAMF_CUCI>(a, cfg, tail_off, n_tail, slot, s);
    }
    return hipErrorInvalidValue;
}
template hipError_t launch_tail<float>(const SgdArgs<float> &, const LaunchCfg &, const int64_t *, int, int64_t, hipStream_t);
template hipError_t launch_tail<double>(const SgdArgs<double> &, const LaunchCfg &, const int64_t *, int, int64_t, hipStream_t);

template <typename T>
hipError_t launch_level_generic(const SgdArgs<T> &a, const LaunchCfg &cfg, int64_t begin, int count, int64_t slot0,
                                hipStream_t s) {
    if (count <= 0) return hipSuccess;
    switch (cfg.model) {
    case BIASEDMF: return launch_generic_model<T, BIASEDMF>(a, cfg, begin, count, slot0, s);
    case PMF: return launch_generic_model<T, PMF>(a, cfg, begin, count, slot0, s);
    case CAMF_C: return launch_generic_model<T, CAMF_C>(a, cfg, begin, count, slot0, s);
    case CAMF_CI: return launch_generic_model<T, CAMF_CI>(a, cfg, begin, count, slot0, s);
    case CAMF_CU: return launch_generic_model<T, CAMF_CU>(a, cfg, begin, count, slot0, s);
    case CAMF_CUCI: return launch_generic_model<T, CAMF_CUCI>(a, cfg, begin, count, slot0, s);
    }
    return hipErrorInvalidValue;
}
template hipError_t launch_level_generic<float>(const SgdArgs<float> &, const LaunchCfg &, int64_t, int, int64_t,
                                                hipStream_t);
template hipError_t launch_level_generic<double>(const SgdArgs<double> &, const LaunchCfg &, int64_t, int, int64_t,
                                                 hipStream_t);

// ---------------------------------------------------------------------------------------------
// CAMF_C, exact and (partly) parallel: conflict-free CRS blocks
// ---------------------------------------------------------------------------------------------
// condBias is read and written by EVERY rating, so CAMF_C's tuples form one dependent chain in CRS order.  But the
// chain only runs through a scalar: with  base_t = ((gm + bu) + bj) + <P[u],Q[j]>  the tuple needs
//     e_t = r_t - (base_t + b[c_1] + ... + b[c_D]),      b[c_d] += lr * (e_t - regC * b[c_d]),
// and everything else (bu, bj, P[u], Q[j]) is an update scaled by e_t.  Inside a run of consecutive CRS tuples that share
// no user and no item ("block", <= 64 tuples, found on the host) every base_t depends only on state from before the
// block, so:   A  all base_t in parallel (gather + dot, rows stay in registers),
//              B  one wave walks the block in CRS order doing only the scalar condBias recurrence (condBias lives in LDS),
//              C  all row / bias updates in parallel from the registers.
// Same values as the sequential loop (no algebraic re-association across tuples); only the dot's summation tree differs,
// as in every non-strict kernel.  One 1024-thread workgroup = 64 sixteen-lane groups walks all blocks of the epoch.
// Pays off when the CRS order is not sorted by user/item (the reference's DataTransformer emits HashMap order); short
// blocks fall back to sgd_serial_fast.
template <int W>
__device__ __forceinline__ double group_sum_t(double x) {
#pragma unroll
    for (int m = W / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}
template <int W>
__device__ __forceinline__ float group_sum_t(float x) { return group_sum<W>(x); }

// One link of CAMF_C's condBias chain costs what its dependent instructions cost on ONE wave: about 8-10 cycles per dependent VALU
// operation, 20 per v_readlane and per taken branch (tools/micro/one_wave_clock.hip: 2.39 GHz, 40 cycles for two dependent VALU ops + the
// loop branch).  So the link is written with as few of each as the arithmetic allows: DM (the number of context dimensions) is a template
// parameter -- no inner loop --, absent conditions (0xff) contribute an exact +0, and the loss terms are left in lanes (ve, vbs) and summed
// once per block.
template <typename T, int DM>
__device__ __forceinline__ void camfc_rc_chain(const int cnt, const int tid, const T vbase, const T my_r, const unsigned long long my_pc,
                                               T &bcreg, T &ve, T &vbs, const T lr, const T regC) {
    const int pc_lo = (int)(unsigned)my_pc, pc_hi = (int)(unsigned)(my_pc >> 32);
    for (int t = 0; t < cnt; ++t) {
        T pred = rl(vbase, t);
        const T rr = rl(my_r, t);
        const unsigned lo = (unsigned)rl(pc_lo, t), hi = DM > 4 ? (unsigned)rl(pc_hi, t) : 0u;
        const T decay = regC * bcreg; // does not depend on this link's error
        T bc_sum = 0;
        bool mine = false;
#pragma unroll
        for (int d = 0; d < DM; ++d) { // the reference adds the deviations one by one, in condition order
            const unsigned cond = ((d < 4 ? lo >> (8 * d) : hi >> (8 * (d - 4))) & 0xffu);
            const bool present = cond != 0xffu;
            const T got = rl(bcreg, (int)(cond & 63u));
            const T v = present ? got : (T)0;
            pred += v;
            bc_sum += v;
            mine = mine || (present && tid == (int)cond);
        }
        const T e = rr - pred;
        if (mine) bcreg = bcreg + lr * (e - decay);
        if (tid == t) {
            ve = e;
            vbs = bc_sum;
        }
    }
}

// The same link with condBias in LDS (any number of conditions; Frappe: 343): lane d reads the entry of the tuple's d-th condition
// (ONE ds_read, 56 cycles of round trip), the entries are taken to the scalar side with DM readlanes issued back to back and added in
// condition order; the tuple's condition ids were staged in LDS by phase A and the next tuple's are read one link ahead.
template <typename T, int DM>
__device__ __forceinline__ void camfc_lds_chain(const int cnt, const int tid, const T vbase, const T my_r, const int32_t *s_conds, T *s_bc,
                                                T &ve, T &vbs, const T lr, const T regC) {
    int cond_next = (tid < DM && cnt > 0) ? s_conds[tid] : -1;
    for (int t = 0; t < cnt; ++t) {
        const int cond = cond_next;
        const int tn = t + 1 < cnt ? t + 1 : t;
        cond_next = tid < DM ? s_conds[tn * DM + tid] : -1;
        const T bc = cond >= 0 ? s_bc[cond] : (T)0;
        T pred = rl(vbase, t);
        const T rr = rl(my_r, t);
        T got[DM];
#pragma unroll
        for (int d = 0; d < DM; ++d) got[d] = rl(bc, d); // an absent condition left 0 in its lane: an exact +0 below
        T bc_sum = 0;
#pragma unroll
        for (int d = 0; d < DM; ++d) { // the reference adds the deviations one by one, in condition order
            pred += got[d];
            bc_sum += got[d];
        }
        const T e = rr - pred;
        if (cond >= 0) s_bc[cond] = bc + lr * (e - regC * bc);
        if (tid == t) {
            ve = e;
            vbs = bc_sum;
        }
    }
}

// RC ("register chain", n_conds <= 64 and dmax <= 8): phase B without an LDS round trip per link -- lane c of wave 0 owns condBias[c],
// lane t holds tuple t's base, rating and its condition ids packed one byte each; a link is readlanes, adds and one masked update
// (0.34 -> 0.1x us per tuple at 28-56 tuples per block, tests/tools/bench_camfc_paths.py).  RC also requests the NEXT block's tuple ids while
// this block's updates are written, so phase A starts with the row gather instead of a dependent id load.
template <typename T, int NV, int CH> // CH: 0 = the round-2 LDS chain, 1 = register chain (RC), 2 = lean LDS chain (any n_conds, dmax <= 8)
__global__ __launch_bounds__(1024) void sgd_camfc_blocks(SgdArgs<T> a, const int32_t *__restrict__ blk_off, int n_blocks,
                                                         double *loss_out) {
    extern __shared__ unsigned char smem_raw[];
    double *s_loss = reinterpret_cast<double *>(smem_raw);       // [64] per-group partials at the end
    T *s_base = reinterpret_cast<T *>(s_loss + 64);               // [64] base_t, then e_t
    T *s_r = s_base + 64;                                         // [64]
    int32_t *s_conds = reinterpret_cast<int32_t *>(s_r + 64);     // [64 x dmax]
    T *s_bc = reinterpret_cast<T *>(s_conds + 64 * (a.dmax > 0 ? a.dmax : 1)); // [n_conds] condBias, resident for the whole epoch
    const int tid = threadIdx.x, l16 = tid & 15, g = tid >> 4;
    const int k = a.k, dmax = a.dmax;
    const HParams hp = *a.hp;
    const T lr = (T)hp.lr, regU = (T)hp.regU, regI = (T)hp.regI, regB = (T)hp.regB, regC = (T)hp.regC, gm = (T)hp.gm;
    for (int c = tid; c < a.n_conds; c += 1024) s_bc[c] = a.condBias[c];
    constexpr bool RC = CH == 1, AHEAD = CH >= 1;
    T bcreg = (RC && tid < a.n_conds) ? a.condBias[tid] : (T)0; // RC: wave 0, lane c = condBias[c]
    double gloss = 0.0;  // groups: e^2-free parts (biases, factors); wave 0 lane 0: e^2 and the condBias term
    int b0 = blk_off[0];
    // RC: ids of the block about to run, requested one block ahead (group g: its tuple's user / item; wave 0 lane t: rating + conditions)
    int uu_n = 0, jj_n = 0;
    T r_n = 0;
    unsigned long long pc_n = 0;
    auto request_ids = [&](int lo, int hi) {
        if (g < hi - lo) {
            uu_n = a.su[(int64_t)lo + g];
            jj_n = a.sj[(int64_t)lo + g];
        }
        if (tid < hi - lo) {
            const int64_t t = (int64_t)lo + tid;
            r_n = a.sr[t];
            if (RC) {
                unsigned long long w = 0;
                for (int d = 0; d < dmax; ++d) {
                    const int c = a.sconds[t * dmax + d];
                    w |= (unsigned long long)(c < 0 ? 0xff : (c & 0xff)) << (8 * d);
                }
                pc_n = w;
            }
        }
    };
    if (AHEAD && n_blocks > 0) request_ids(b0, blk_off[1]);
    __syncthreads();
    for (int blk = 0; blk < n_blocks; ++blk) {
        const int b1 = blk_off[blk + 1];
        const int cnt = b1 - b0; // <= 64: tuple g of the block belongs to group g
        // ---- A: base_t, rows into registers
        const bool live = g < cnt;
        int uu = 0, jj = 0;
        T p[NV], q[NV], bu = 0, bj = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) p[i] = q[i] = 0;
        const T my_r = r_n;                    // RC, wave 0: lane t = tuple t of this block
        const unsigned long long my_pc = pc_n;
        if (live) {
            const int64_t t = (int64_t)b0 + g;
            if (AHEAD) {
                uu = uu_n;
                jj = jj_n;
                if (!RC && l16 < dmax) s_conds[g * dmax + l16] = a.sconds[t * dmax + l16];
            } else {
                uu = a.su[t];
                jj = a.sj[t];
                if (l16 < dmax) s_conds[g * dmax + l16] = a.sconds[t * dmax + l16];
                if (l16 == 0) s_r[g] = a.sr[t];
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int f = l16 + 16 * i;
                if (f < k) {
                    p[i] = a.P[(size_t)uu * k + f];
                    q[i] = a.Q[(size_t)jj * k + f];
                }
            }
            bu = a.userBias[uu];
            bj = a.itemBias[jj];
        }
        // the NEXT block's ids: requested now, a whole block (gather, chain, updates) before they are used
        if (AHEAD && blk + 1 < n_blocks) request_ids(b1, blk_off[blk + 2]);
        T part = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) part += p[i] * q[i];
        const T dot = group_sum_t<16>(part);
        if (live && l16 == 0) s_base[g] = ((gm + bu) + bj) + dot;
        __syncthreads();
        // ---- B: the scalar chain, wave 0, CRS order
        if (RC) {
            if (tid < 64) {
                const T vbase = tid < cnt ? s_base[tid] : (T)0;
                T ve = 0, vbs = 0; // lane t: e_t and the plain sum of tuple t's condBias entries (reference quirk, CAMF_C.java:110,115)
                switch (dmax) {
                case 1: camfc_rc_chain<T, 1>(cnt, tid, vbase, my_r, my_pc, bcreg, ve, vbs, lr, regC); break;
                case 2: camfc_rc_chain<T, 2>(cnt, tid, vbase, my_r, my_pc, bcreg, ve, vbs, lr, regC); break;
                case 3: camfc_rc_chain<T, 3>(cnt, tid, vbase, my_r, my_pc, bcreg, ve, vbs, lr, regC); break;
                case 4: camfc_rc_chain<T, 4>(cnt, tid, vbase, my_r, my_pc, bcreg, ve, vbs, lr, regC); break;
                case 5: camfc_rc_chain<T, 5>(cnt, tid, vbase, my_r, my_pc, bcreg, ve, vbs, lr, regC); break;
                case 6: camfc_rc_chain<T, 6>(cnt, tid, vbase, my_r, my_pc, bcreg, ve, vbs, lr, regC); break;
                case 7: camfc_rc_chain<T, 7>(cnt, tid, vbase, my_r, my_pc, bcreg, ve, vbs, lr, regC); break;
                default: camfc_rc_chain<T, 8>(cnt, tid, vbase, my_r, my_pc, bcreg, ve, vbs, lr, regC); break;
                }
                if (tid < cnt) s_base[tid] = ve; // base_t is consumed: the slot now carries e_t for phase C
                // the block's loss terms, once per block instead of once per link
                const double lsum = wave_sum_dpp(tid < cnt ? (double)(ve * ve) + (double)(regB * vbs) : 0.0);
                if (tid == 0) gloss += lsum;
            }
        } else if (CH == 2) {
            if (tid < 64) {
                const T vbase = tid < cnt ? s_base[tid] : (T)0;
                T ve = 0, vbs = 0;
                switch (dmax) {
                case 1: camfc_lds_chain<T, 1>(cnt, tid, vbase, my_r, s_conds, s_bc, ve, vbs, lr, regC); break;
                case 2: camfc_lds_chain<T, 2>(cnt, tid, vbase, my_r, s_conds, s_bc, ve, vbs, lr, regC); break;
                case 3: camfc_lds_chain<T, 3>(cnt, tid, vbase, my_r, s_conds, s_bc, ve, vbs, lr, regC); break;
                case 4: camfc_lds_chain<T, 4>(cnt, tid, vbase, my_r, s_conds, s_bc, ve, vbs, lr, regC); break;
                case 5: camfc_lds_chain<T, 5>(cnt, tid, vbase, my_r, s_conds, s_bc, ve, vbs, lr, regC); break;
                case 6: camfc_lds_chain<T, 6>(cnt, tid, vbase, my_r, s_conds, s_bc, ve, vbs, lr, regC); break;
                case 7: camfc_lds_chain<T, 7>(cnt, tid, vbase, my_r, s_conds, s_bc, ve, vbs, lr, regC); break;
                default: camfc_lds_chain<T, 8>(cnt, tid, vbase, my_r, s_conds, s_bc, ve, vbs, lr, regC); break;
                }
                if (tid < cnt) s_base[tid] = ve;
                const double lsum = wave_sum_dpp(tid < cnt ? (double)(ve * ve) + (double)(regB * vbs) : 0.0);
                if (tid == 0) gloss += lsum;
            }
        } else
        // (round-2 LDS chain) lane d owns the tuple's d-th condition
        if (tid < 64) {
            double l = 0.0;
            for (int t = 0; t < cnt; ++t) {
                int cond = -1;
                T bc = 0;
                if (tid < dmax) cond = s_conds[t * dmax + tid];
                if (cond >= 0) bc = s_bc[cond];
                T pred = s_base[t];
                T bc_sum = 0;
                const unsigned long long present = __ballot(cond >= 0);
                for (int d = 0; d < dmax; ++d) // the reference adds the deviations one by one, in condition order
                    if ((present >> d) & 1ull) {
                        const T v = rl(bc, d);
                        pred += v;
                        bc_sum += v; // plain sum, weighted by regB: reference quirk (CAMF_C.java:110,115)
                    }
                const T e = s_r[t] - pred;
                if (cond >= 0) s_bc[cond] = bc + lr * (e - regC * bc);
                if (tid == 0) s_base[t] = e; // base_t is consumed: the slot now carries e_t for phase C
                l += (double)(e * e) + (double)(regB * bc_sum);
            }
            if (tid == 0) gloss += l;
        }
        __syncthreads();
        // ---- C: everything scaled by e_t, from the registers
        if (live) {
            const T e = s_base[g];
            T reg_part = 0;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int f = l16 + 16 * i;
                if (f < k) {
                    const T pv = p[i], qv = q[i];
                    a.P[(size_t)uu * k + f] = pv + lr * (e * qv - regU * pv);
                    a.Q[(size_t)jj * k + f] = qv + lr * (e * pv - regI * qv);
                    reg_part += (regU * pv) * pv + (regI * qv) * qv;
                }
            }
            const T reg_sum = group_sum_t<16>(reg_part);
            if (l16 == 0) {
                a.userBias[uu] = bu + lr * (e - regB * bu);
                a.itemBias[jj] = bj + lr * (e - regB * bj);
                gloss += (double)((regB * bu) * bu) + (double)((regB * bj) * bj) + (double)reg_sum;
            }
        }
        b0 = b1;
        __syncthreads(); // this block's rows are visible (workgroup scope) before the next block gathers
    }
    if (l16 == 0) s_loss[g] = gloss;
    __syncthreads();
    if (RC) {
        if (tid < a.n_conds) a.condBias[tid] = bcreg;
    } else
        for (int c = tid; c < a.n_conds; c += 1024) a.condBias[c] = s_bc[c];
    if (tid == 0) {
        double sum = 0.0;
        for (int i = 0; i < 64; ++i) sum += s_loss[i];
        loss_out[0] = sum * 0.5;
    }
}

size_t camfc_blocks_lds(int n_conds, int dmax, size_t esize) {
    return (size_t)n_conds * esize + 128 * esize + 64 * sizeof(double) + (size_t)64 * (dmax > 0 ? dmax : 1) * sizeof(int32_t) + 64;
}

template <typename T>
hipError_t launch_camfc_blocks(const SgdArgs<T> &a, const int32_t *blk_off, int n_blocks, double *loss_out, hipStream_t s) {
    const size_t lds = camfc_blocks_lds(a.n_conds, a.dmax, sizeof(T));
    // chain form: 1 = condBias in a register (<= 64 conditions), 2 = lean LDS chain (any number), 0 = the round-2 LDS chain (dmax > 8, or
    // CMI_CAMFC_LDS_CHAIN=1 for A/B runs)
    const int ch = (a.dmax > 8 || a.dmax < 1 || getenv("CMI_CAMFC_LDS_CHAIN")) ? 0 : (a.n_conds <= 64 && !getenv("CMI_CAMFC_NO_RC")) ? 1 : 2;
#define CMI_CAMFC_LAUNCH(NV)                                                                                                              \
    do {                                                                                                                                  \
        if (ch == 1) hipLaunchKernelGGL((sgd_camfc_blocks<T, NV, 1>), dim3(1), dim3(1024), lds, s, a, blk_off, n_blocks, loss_out);      \
        else if (ch == 2) hipLaunchKernelGGL((sgd_camfc_blocks<T, NV, 2>), dim3(1), dim3(1024), lds, s, a, blk_off, n_blocks, loss_out); \
        else hipLaunchKernelGGL((sgd_camfc_blocks<T, NV, 0>), dim3(1), dim3(1024), lds, s, a, blk_off, n_blocks, loss_out);              \
    } while (0)
    if (a.k <= 64) CMI_CAMFC_LAUNCH(4);
    else if (a.k <= 128) CMI_CAMFC_LAUNCH(8);
    else CMI_CAMFC_LAUNCH(16);
#undef CMI_CAMFC_LAUNCH
    return hipGetLastError();
}
template hipError_t launch_camfc_blocks<float>(const SgdArgs<float> &, const int32_t *, int, double *, hipStream_t);
template hipError_t launch_camfc_blocks<double>(const SgdArgs<double> &, const int32_t *, int, double *, hipStream_t);

template <typename T, int MODEL>
static hipError_t launch_serial_model(const SgdArgs<T> &a, const LaunchCfg &cfg, int64_t n, double *loss_out,
                                      hipStream_t s) {
    const size_t lds = (MODEL == CAMF_C ? (size_t)a.n_conds * sizeof(T) : 0) + (size_t)64 * a.dmax * sizeof(int32_t) + 16;
    if (cfg.strict)
        hipLaunchKernelGGL((sgd_serial<T, MODEL, true>), dim3(1), dim3(64), 0, s, a, n, loss_out);
    else if (MODEL == CAMF_C && camfc_pipe_supported(a.k, a.n_conds, a.dmax))
        return launch_camfc_pipe<T>(a, n, loss_out, s);
    else if (a.k <= 256 && lds <= 64 * 1024 && !cmi_exp_env("CMI_SERIAL_GENERIC")) {
        if (a.k == 64) hipLaunchKernelGGL((sgd_serial_fast<T, MODEL, 1, true>), dim3(1), dim3(64), lds, s, a, n, loss_out);
        else if (a.k == 128) hipLaunchKernelGGL((sgd_serial_fast<T, MODEL, 2, true>), dim3(1), dim3(64), lds, s, a, n, loss_out);
        else if (a.k == 256) hipLaunchKernelGGL((sgd_serial_fast<T, MODEL, 4, true>), dim3(1), dim3(64), lds, s, a, n, loss_out);
        else if (a.k < 64) hipLaunchKernelGGL((sgd_serial_fast<T, MODEL, 1, false>), dim3(1), dim3(64), lds, s, a, n, loss_out);
        else if (a.k < 128) hipLaunchKernelGGL((sgd_serial_fast<T, MODEL, 2, false>), dim3(1), dim3(64), lds, s, a, n, loss_out);
        else hipLaunchKernelGGL((sgd_serial_fast<T, MODEL, 4, false>), dim3(1), dim3(64), lds, s, a, n, loss_out);
    }
    else
        hipLaunchKernelGGL((sgd_serial<T, MODEL, false>), dim3(1), dim3(64), 0, s, a, n, loss_out);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_serial(const SgdArgs<T> &a, const LaunchCfg &cfg, int64_t n, double *loss_out, hipStream_t s) {
    switch (cfg.model) {
    case BIASEDMF: return launch_serial_model<T, BIASEDMF>(a, cfg, n, loss_out, s);
    case PMF: return launch_serial_model<T, PMF>(a, cfg, n, loss_out, s);
    case CAMF_C: return launch_serial_model<T, CAMF_C>(a, cfg, n, loss_out, s);
    case CAMF_CI: return launch_serial_model<T, CAMF_CI>(a, cfg, n, loss_out, s);
    case CAMF_CU: return launch_serial_model<T, CAMF_CU>(a, cfg, n, loss_out, s);
    case CAMF_CUCI: return launch_serial_model<T, CAMF_CUCI>(a, cfg, n, loss_out, s);
    }
    return hipErrorInvalidValue;
}
template hipError_t launch_serial<float>(const SgdArgs<float> &, const LaunchCfg &, int64_t, double *, hipStream_t);
template hipError_t launch_serial<double>(const SgdArgs<double> &, const LaunchCfg &, int64_t, double *, hipStream_t);

hipError_t launch_set_hparams(HParams *dst, HParams v, hipStream_t s) {
    hipLaunchKernelGGL(set_hparams_kernel, dim3(1), dim3(1), 0, s, dst, v);
    return hipGetLastError();
}

hipError_t launch_reduce_loss(const double *loss_part, int64_t n_slots, double *scratch, double *loss_out,
                              hipStream_t s) {
    int nblk = (int)((n_slots + 4095) / 4096);
    if (nblk < 1) nblk = 1;
    if (nblk > 256) nblk = 256;
    hipLaunchKernelGGL(reduce_loss_stage1, dim3(nblk), dim3(256), 0, s, loss_part, n_slots, scratch);
    hipLaunchKernelGGL(reduce_loss_stage2, dim3(1), dim3(256), 0, s, scratch, nblk, loss_out);
    return hipGetLastError();
}

int eval_blocks(int64_t n) {
    int64_t b = (n + 3) / 4;
    if (b < 1) b = 1;
    if (b > 4096) b = 4096;
    return (int)b;
}

template <typename T>
hipError_t launch_eval(const EvalArgs<T> &a, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(eval_kernel<T>, dim3(eval_blocks(n)), dim3(256), 0, s, a, n);
    return hipGetLastError();
}
template hipError_t launch_eval<float>(const EvalArgs<float> &, int64_t, hipStream_t);
template hipError_t launch_eval<double>(const EvalArgs<double> &, int64_t, hipStream_t);

static unsigned elementwise_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
hipError_t launch_delta_pack(const void *state, const void *snap, void *bucket, int64_t n, bool f64, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (f64) hipLaunchKernelGGL(delta_pack_kernel<double>, dim3(elementwise_blocks(n)), dim3(256), 0, s, (const double *)state, (const double *)snap, (double *)bucket, n);
    else if (n % 4 == 0 && ((uintptr_t)state | (uintptr_t)snap | (uintptr_t)bucket) % 16 == 0)
        hipLaunchKernelGGL(delta_pack_f32x4, dim3(elementwise_blocks(n / 4)), dim3(256), 0, s, (const float4 *)state, (const float4 *)snap, (float4 *)bucket, n / 4);
    else hipLaunchKernelGGL(delta_pack_kernel<float>, dim3(elementwise_blocks(n)), dim3(256), 0, s, (const float *)state, (const float *)snap, (float *)bucket, n);
    return hipGetLastError();
}
hipError_t launch_delta_apply(void *state, void *snap, const void *bucket, double scale, int64_t n, bool f64, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (f64) hipLaunchKernelGGL(delta_apply_kernel<double>, dim3(elementwise_blocks(n)), dim3(256), 0, s, (double *)state, (double *)snap, (const double *)bucket, scale, n);
    else if (n % 4 == 0 && ((uintptr_t)state | (uintptr_t)snap | (uintptr_t)bucket) % 16 == 0)
        hipLaunchKernelGGL(delta_apply_f32x4, dim3(elementwise_blocks(n / 4)), dim3(256), 0, s, (float4 *)state, (float4 *)snap, (const float4 *)bucket, (float)scale, n / 4);
    else hipLaunchKernelGGL(delta_apply_kernel<float>, dim3(elementwise_blocks(n)), dim3(256), 0, s, (float *)state, (float *)snap, (const float *)bucket, (float)scale, n);
    return hipGetLastError();
}

hipError_t launch_convert(const void *src, int src_f64, void *dst, int dst_f64, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const dim3 grid((unsigned)blocks), block(256);
    if (src_f64 && dst_f64)
        hipLaunchKernelGGL((convert_kernel<double, double>), grid, block, 0, s, (const double *)src, (double *)dst, n);
    else if (src_f64 && !dst_f64)
        hipLaunchKernelGGL((convert_kernel<double, float>), grid, block, 0, s, (const double *)src, (float *)dst, n);
    else if (!src_f64 && dst_f64)
        hipLaunchKernelGGL((convert_kernel<float, double>), grid, block, 0, s, (const float *)src, (double *)dst, n);
    else
        hipLaunchKernelGGL((convert_kernel<float, float>), grid, block, 0, s, (const float *)src, (float *)dst, n);
    return hipGetLastError();
}

} // namespace cmi

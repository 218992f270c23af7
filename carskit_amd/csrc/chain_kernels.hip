// chain_kernels.hip -- the hub-chain level kernel (gfx950): the default SGD path for BiasedMF / PMF / CAMF_CI / CAMF_CU /
// CAMF_CUCI whenever the schedule's levels are wide (reference loop: e.g. CAMF_CI.java:79-123).
//
// The plain level kernel (mf_sgd_kernels.hip) reads and writes BOTH rows of every tuple from HBM and needs one launch per
// tuple of the busiest row.  Here a 16-lane group walks a UNIT of the chain schedule (level_schedule.cpp,
// build_chain_schedule): up to max_chain tuples that are consecutive in the CRS order of one HUB row.  The hub row
// (Q[j] when chaining along items, P[u] along users), its scalar bias and its context-bias row stay ON CHIP for the whole
// unit -- factors in registers, the icBias / ucBias row in LDS -- and only the SPOKE rows stream through: the hub side's
// HBM traffic drops by the mean unit length and the epoch has as many launches as the longest chain of UNITS.
// Per tuple the arithmetic is the same expression, in the same order, as fast_tuples_f32 -- so for fp32 state the model is
// bit-identical to the plain level schedule's (tests/test_gpu_chain.py) -- and T = double gives the fp64-state fast path.
//
// Lane layout as in sgd_level_fast_f32: lane l of the group owns the 16-byte vectors {l, l+16, ...} of both rows (whole
// 256-B segments per load/store instruction), lane d < D owns the tuple's d-th condition, lane 0 the scalar biases; dot and
// loss partials are reduced with DPP row rotations.  Spoke rows are prefetched one tuple ahead, tuple ids two ahead.
// HBM-bound gather/scatter: no MFMA.
#include "mf_sgd_kernels.hpp"
#include "sgd_device.hpp"

#include <cstdlib>

namespace cmi {

template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
    typedef float4 type;
    static constexpr int E = 4;
};
template <>
struct Vec16<double> {
    typedef double2 type;
    static constexpr int E = 2;
};
__device__ __forceinline__ void unpack(const float4 &v, float *x) { x[0] = v.x, x[1] = v.y, x[2] = v.z, x[3] = v.w; }
__device__ __forceinline__ void unpack(const double2 &v, double *x) { x[0] = v.x, x[1] = v.y; }
__device__ __forceinline__ float4 pack(const float *x) { return make_float4(x[0], x[1], x[2], x[3]); }
__device__ __forceinline__ double2 pack(const double *x) { return make_double2(x[0], x[1]); }

// The hub's context-bias row lives in LDS and is read / written by DIFFERENT lanes of the same wave from one tuple to the next.
// LDS operations of a wave execute in program order, so all that is needed is that the compiler keeps that order: a
// compiler-only barrier (a fence would also drain the spoke-row prefetch).
__device__ __forceinline__ void chain_lds_order() { asm volatile("" ::: "memory"); }

// the spoke side of one tuple: factor vectors, scalar bias (lane-uniform), the lane's context-bias cell
template <typename T, int NV>
struct SpokeRow {
    typename Vec16<T>::type v[NV];
    T sc;
    T *psc;
    int spoke, cond;
    T *dst; // where the updated row goes: the model table's row, or (spoke arena) the slot of the row's next tuple
};

// pos / nx: stream position of the tuple and of the same spoke row's next tuple (only read when the spoke arena is on)
template <typename T, int MODEL, int NV, bool RAGGED, bool HUB_ITEM>
__device__ __forceinline__ void chain_load_spoke(const SgdArgs<T> &a, int spoke, int cond, int l16, int K, SpokeRow<T, NV> &r, int64_t pos = 0,
                                                 int nx = 0) {
    using M = Traits<MODEL>;
    using V = typename Vec16<T>::type;
    constexpr int E = Vec16<T>::E;
    constexpr bool SC = HUB_ITEM ? M::has_uc : M::has_ic;  // context-bias table on the spoke side (the scalar bias: chain_unit)
    T *tab = HUB_ITEM ? a.P : a.Q;
    // Spoke arena (a.arena != null, wave-uniform): the row of the tuple at stream position `pos` sits in arena[pos] -- a unit's rows are
    // CONTIGUOUS there and units follow each other in launch order, so the spoke reads of a level are one sequential stream instead of
    // random 512-B rows over a multi-GB table (north_star: 2.4 address-translation misses per tuple).  The updated
    // row is written to the slot of the same row's NEXT tuple: random full-line writes.
    const T *src = a.arena ? a.arena + (size_t)pos * K : tab + (size_t)spoke * K;
    r.dst = a.arena ? a.arena + (size_t)nx * K : tab + (size_t)spoke * K;
    const V *row = reinterpret_cast<const V *>(src) + l16;
    r.spoke = spoke;
    r.cond = cond;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        T z[E];
#pragma unroll
        for (int c = 0; c < E; ++c) z[c] = (T)0;
        r.v[v] = pack(z);
        if (!RAGGED || E * l16 + 16 * E * v < K) r.v[v] = row[v * 16];
    }
    r.sc = (T)0;
    r.psc = nullptr;
    if (SC && cond >= 0) {
        r.psc = (HUB_ITEM ? a.ucBias : a.icBias) + (size_t)spoke * a.n_conds + cond;
        r.sc = *r.psc;
    }
}

template <typename T>
struct ChainHp {
    T lr, regU, regI, regB, regC, gm;
};

// One update: hub registers (h, hb, the LDS copy of the hub's context-bias row) x spoke row `cur`.  p = user side, q = item side;
// every expression is the one of fast_tuples_f32 (mf_sgd_kernels.hip), in the same order.
template <typename T, int MODEL, int NV, bool RAGGED, bool HUB_ITEM>
__device__ __forceinline__ void chain_step(const SgdArgs<T> &a, const ChainHp<T> &hp, T (&h)[NV][Vec16<T>::E], T &hb, T *s_hc, T *s_sb,
                                           const SpokeRow<T, NV> &cur, T rr, int l16, int K, double &gloss) {
    using M = Traits<MODEL>;
    using V = typename Vec16<T>::type;
    constexpr int E = Vec16<T>::E;
    constexpr bool HB = HUB_ITEM ? M::has_bj : M::has_bu;  // scalar bias on the hub side
    constexpr bool SB = HUB_ITEM ? M::has_bu : M::has_bj;
    constexpr bool HC = HUB_ITEM ? M::has_ic : M::has_uc;  // context-bias table on the hub side (row kept in LDS)
    constexpr bool SC = HUB_ITEM ? M::has_uc : M::has_ic;
    const T lr = hp.lr, regU = hp.regU, regI = hp.regI, regB = hp.regB, regC = hp.regC;

    T sp[NV][E];
#pragma unroll
    for (int v = 0; v < NV; ++v) unpack(cur.v[v], sp[v]);
    T part = (T)0;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int c = 0; c < E; ++c) part += HUB_ITEM ? sp[v][c] * h[v][c] : h[v][c] * sp[v][c];
    const T dot = row_sum16(part);

    T hcv = (T)0; // the lane's hub-side context-bias cell (from the LDS copy of the hub's row)
    if (HC) {
        chain_lds_order();
        if (cur.cond >= 0) hcv = s_hc[cur.cond];
    }
    const T sbv = SB ? *s_sb : (T)0; // the spoke's scalar bias: this tuple's slot of the unit's bias batch (LDS)
    const T bu = HUB_ITEM ? sbv : hb, bj = HUB_ITEM ? hb : sbv;
    const T bic = HUB_ITEM ? hcv : cur.sc, buc = HUB_ITEM ? cur.sc : hcv;
    T pred = hp.gm;
    if (M::has_bu) pred += bu;
    if (M::has_bj) pred += bj;
    pred += dot;
    if (M::has_ctx) {
        T term = (T)0; // lane d carries the deviation of the tuple's d-th condition
        if (M::has_ic && M::has_uc) term = bic + buc;
        else if (M::has_ic) term = bic;
        else if (M::has_uc) term = buc;
        pred += row_sum16(term);
    }
    const T e = rr - pred;

    // scalar biases: the hub's stays in registers, the spoke's goes back to its LDS slot (the unit's biases leave in one store)
    if (HB) hb = hb + lr * (e - regB * hb);
    if (SB && l16 == 0) *s_sb = sbv + lr * (e - regB * sbv);
    T ctx_loss = (T)0;
    if (cur.cond >= 0) {
        if (M::has_ic) ctx_loss += bic * bic;
        if (M::has_uc) ctx_loss += buc * buc;
        if (HC) s_hc[cur.cond] = hcv + lr * (e - regC * hcv);
        if (SC) *cur.psc = cur.sc + lr * (e - regC * cur.sc);
    }
    if (HC) chain_lds_order();

    T lsum = (T)0;
    V *srow = reinterpret_cast<V *>(cur.dst) + l16;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        T sn[E];
#pragma unroll
        for (int c = 0; c < E; ++c) {
            const T p = HUB_ITEM ? sp[v][c] : h[v][c], q = HUB_ITEM ? h[v][c] : sp[v][c];
            const T pn = p + lr * (e * q - regU * p);
            const T qn = q + lr * (e * p - regI * q);
            lsum += (regU * p) * p + (regI * q) * q;
            h[v][c] = HUB_ITEM ? qn : pn;
            sn[c] = HUB_ITEM ? pn : qn;
        }
        if (!RAGGED || E * l16 + 16 * E * v < K) srow[v * 16] = pack(sn);
    }
    const T reg_loss = row_sum16(lsum);
    const T ctx_sum = M::has_ctx ? row_sum16(ctx_loss) : (T)0;
    if (l16 == 0) {
        double l = (double)e * (double)e;
        if (M::has_bu) l += (double)regB * bu * bu;
        if (M::has_bj) l += (double)regB * bj * bj;
        if (M::has_ctx) l += (double)regC * ctx_sum;
        gloss += l + (double)reg_loss;
    }
}

// LDS per 16-lane group: [hub context-bias row: n_conds x T (if the hub side has one)] [ratings: 16 x T] [spoke scalar biases: 16 x T]
// [spoke ids: 16 x i32]
// [next positions: 16 x i32 (spoke arena)]
// [condition ids: 16 x dmax x i32] -- the unit's ids are staged once (one coalesced round trip) so that the chain loop has no
// dependent global id -> row load pairs.
__host__ __device__ inline size_t chain_group_lds(int n_conds_hub, int dmax, size_t esize) {
    size_t b = (size_t)n_conds_hub * esize + 32 * esize + 32 * 4 + (size_t)16 * (dmax > 0 ? dmax : 0) * 4;
    return (b + 15) & ~(size_t)15;
}

// Everything a group needs to know about its unit before it touches the model: bounds and ids (none of it depends on the
// updates of earlier levels, so a multi-level walk loads it one level ahead).
template <typename T>
struct ChainPre {
    int32_t tb, len;
    int hub, sp0, cd0, my_sp, my_nx, nx0;
    T my_rr;
    int cdv[4];
};

// round trip 2 of a unit: every id load is issued before the first one is used
template <typename T, bool HAS_CTX, bool HUB_ITEM>
__device__ __forceinline__ ChainPre<T> chain_prefetch_ids(const SgdArgs<T> &a, int32_t tb, int32_t te, int l16, int dmax) {
    ChainPre<T> p;
    p.tb = tb;
    p.len = te - tb; // 1 .. 16
    p.hub = HUB_ITEM ? a.sj[tb] : a.su[tb];
    p.sp0 = HUB_ITEM ? a.su[tb] : a.sj[tb];
    p.cd0 = (HAS_CTX && l16 < dmax) ? a.sconds[(int64_t)tb * dmax + l16] : -1;
    p.my_sp = 0;
    p.my_nx = 0;
    p.my_rr = (T)0;
    p.nx0 = a.next_pos ? a.next_pos[tb] : 0;
    if (l16 < p.len) {
        p.my_sp = HUB_ITEM ? a.su[tb + l16] : a.sj[tb + l16];
        p.my_rr = a.sr[tb + l16];
        if (a.next_pos) p.my_nx = a.next_pos[tb + l16];
    }
    const int n_cd = p.len * dmax; // <= 256 condition ids, staged 16 per pass
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        p.cdv[r] = -1;
        if (l16 + 16 * r < n_cd) p.cdv[r] = a.sconds[(int64_t)tb * dmax + l16 + 16 * r];
    }
    return p;
}

// One unit: round trip 3 (hub row + first spoke row), the chain, the write-back of the hub side.  gbase = this group's LDS.
template <typename T, int MODEL, int NV, bool RAGGED, bool HUB_ITEM>
__device__ __forceinline__ void chain_unit(const SgdArgs<T> &a, const ChainHp<T> &hp, const ChainPre<T> &pre, unsigned char *gbase, int l16,
                                           int K, int dmax, double &gloss) {
    using M = Traits<MODEL>;
    using V = typename Vec16<T>::type;
    constexpr int E = Vec16<T>::E;
    constexpr bool HB = HUB_ITEM ? M::has_bj : M::has_bu;
    constexpr bool HC = HUB_ITEM ? M::has_ic : M::has_uc;
    constexpr bool SB = HUB_ITEM ? M::has_bu : M::has_bj;
    const int len = pre.len;
    const int32_t tb = pre.tb;
    T *s_hc = reinterpret_cast<T *>(gbase);
    T *s_rr = s_hc + (HC ? a.n_conds : 0);
    T *s_sb = s_rr + 16;
    int *s_sp = reinterpret_cast<int *>(s_sb + 16);
    int *s_nx = s_sp + 16;
    int *s_cd = s_nx + 16;
    const int n_cd = len * dmax;

    // ---- round trip 3: the hub row comes on chip once, together with the first spoke row
    T *htab = HUB_ITEM ? a.Q : a.P;
    V *hrow = reinterpret_cast<V *>(htab + (size_t)pre.hub * K) + l16;
    T h[NV][E];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int c = 0; c < E; ++c) h[v][c] = (T)0;
        if (!RAGGED || E * l16 + 16 * E * v < K) {
            const V t = hrow[v * 16];
            unpack(t, h[v]);
        }
    }
    T hb = (T)0;
    T *phb = nullptr;
    if (HB) {
        phb = (HUB_ITEM ? a.itemBias : a.userBias) + pre.hub;
        hb = *phb;
    }
    T *hc_row = nullptr;
    T hcv4[4];
    if (HC) {
        hc_row = (HUB_ITEM ? a.icBias : a.ucBias) + (size_t)pre.hub * a.n_conds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hcv4[r] = (T)0;
            if (l16 + 16 * r < a.n_conds) hcv4[r] = hc_row[l16 + 16 * r];
        }
    }
    // the scalar biases of ALL the unit's spoke rows in one load (lane i: tuple i's; a unit's spoke rows are pairwise distinct) instead
    // of one 4-byte gather and one 4-byte scatter per tuple, each a memory instruction with one useful lane in sixteen
    T *sb_tab = HUB_ITEM ? a.userBias : a.itemBias;
    T my_sb = (T)0;
    if (SB && l16 < len) my_sb = sb_tab[pre.my_sp];
    SpokeRow<T, NV> A, B;
    chain_load_spoke<T, MODEL, NV, RAGGED, HUB_ITEM>(a, pre.sp0, pre.cd0, l16, K, A, tb, pre.nx0);

    // ---- ids, the spoke biases and the hub's context-bias row into LDS
    if (l16 < len) {
        s_sp[l16] = pre.my_sp;
        s_nx[l16] = pre.my_nx;
        s_rr[l16] = pre.my_rr;
        if (SB) s_sb[l16] = my_sb;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (l16 + 16 * r < n_cd) s_cd[l16 + 16 * r] = pre.cdv[r];
    for (int c = l16 + 64; c < n_cd; c += 16) s_cd[c] = a.sconds[(int64_t)tb * dmax + c]; // dmax > 4 with long units
    if (HC) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (l16 + 16 * r < a.n_conds) s_hc[l16 + 16 * r] = hcv4[r];
        for (int c = l16 + 64; c < a.n_conds; c += 16) s_hc[c] = hc_row[c]; // more than 64 conditions
    }
    chain_lds_order();

    // ---- the chain: spoke rows ping-pong between two register sets, the next one in flight while this one is used
    // The chain.  Spoke rows ping-pong between two register sets: the row of tuple i + 1 is requested before tuple i is computed.
    // What it takes for the waits to stay PARTIAL (round 2's loop compiled to s_waitcnt vmcnt(0) right after the read-ahead's issue,
    // i.e. load -> wait -> compute; `hipcc -S` excerpt in DESIGN.md section 5):
    //  * the read-ahead is issued unconditionally -- behind `if (i + 1 < len)` the compiler cannot count the loads in flight at the
    //    join.  Past the end of the unit the slot re-reads the unit's last row (an L1 / L2 hit, discarded);
    //  * the loop has ONE exit, at the bottom, after a whole pair of tuples: an exit between the two halves leaves a static path
    //    "second half skipped -> loop header" on which the read-ahead's destination registers are still pending, and the header's
    //    address arithmetic reuses them -> a wait at the top of every iteration, ahead of the read-ahead's issue.  An odd tuple runs
    //    after the loop;
    //  * nothing is in flight on entry (the prologue's pending loads would be merged into the header's bookkeeping the same way).
    // Per pair the wave then issues [load B][store A][load A'][store B] and waits with vmcnt(4) / vmcnt(5): the two loads of the row
    // read ahead and the two stores of the previous row stay in flight while a row is computed.
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    const int pairs = len >> 1;
    for (int t = 0; t < pairs; ++t) {
        const int i = 2 * t;
        chain_load_spoke<T, MODEL, NV, RAGGED, HUB_ITEM>(a, s_sp[i + 1], l16 < dmax ? s_cd[(i + 1) * dmax + l16] : -1, l16, K, B, tb + i + 1, s_nx[i + 1]);
        chain_lds_order();
        chain_step<T, MODEL, NV, RAGGED, HUB_ITEM>(a, hp, h, hb, s_hc, s_sb + i, A, s_rr[i], l16, K, gloss);
        {
            const int nx = i + 2 < len ? i + 2 : len - 1;
            chain_load_spoke<T, MODEL, NV, RAGGED, HUB_ITEM>(a, s_sp[nx], l16 < dmax ? s_cd[nx * dmax + l16] : -1, l16, K, A, tb + nx, s_nx[nx]);
        }
        chain_lds_order();
        chain_step<T, MODEL, NV, RAGGED, HUB_ITEM>(a, hp, h, hb, s_hc, s_sb + i + 1, B, s_rr[i + 1], l16, K, gloss);
    }
    if (len & 1) chain_step<T, MODEL, NV, RAGGED, HUB_ITEM>(a, hp, h, hb, s_hc, s_sb + len - 1, A, s_rr[len - 1], l16, K, gloss);

    // ---- the hub row leaves the chip once
#pragma unroll
    for (int v = 0; v < NV; ++v)
        if (!RAGGED || E * l16 + 16 * E * v < K) hrow[v * 16] = pack(h[v]);
    if (HB && l16 == 0) *phb = hb;
    if (SB) { // the unit's spoke biases leave in one store
        chain_lds_order();
        if (l16 < len) sb_tab[s_sp[l16]] = s_sb[l16];
    }
    if (HC) {
        chain_lds_order();
        for (int c = l16; c < a.n_conds; c += 16) hc_row[c] = s_hc[c];
    }
}

template <typename T, int MODEL, int NV, bool RAGGED, bool HUB_ITEM>
__global__ __launch_bounds__(256) void sgd_chain_level(SgdArgs<T> a, const int32_t *__restrict__ unit_off, int64_t ubegin, int count,
                                                       int64_t slot0) {
    using M = Traits<MODEL>;
    constexpr int E = Vec16<T>::E;
    static_assert(MODEL != CAMF_C, "CAMF_C has no level schedule (shared condBias)");
    constexpr bool HC = HUB_ITEM ? M::has_ic : M::has_uc;
    extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
    __shared__ double s_loss[16];
    const int tid = threadIdx.x, l16 = tid & 15, gib = tid >> 4;
    const int g = blockIdx.x * 16 + gib;
    const int K = RAGGED ? a.k : NV * 16 * E;
    const int dmax = M::has_ctx ? a.dmax : 0;
    double gloss = 0.0;
    const HParams hpd = *a.hp;
    const ChainHp<T> hp{(T)hpd.lr, (T)hpd.regU, (T)hpd.regI, (T)hpd.regB, (T)hpd.regC, (T)hpd.gm};

    if (g < count) { // group-uniform
        const int32_t tb = unit_off[ubegin + g], te = unit_off[ubegin + g + 1];  // round trip 1
        const ChainPre<T> pre = chain_prefetch_ids<T, M::has_ctx, HUB_ITEM>(a, tb, te, l16, dmax);
        chain_unit<T, MODEL, NV, RAGGED, HUB_ITEM>(a, hp, pre, chain_smem + (size_t)gib * chain_group_lds(HC ? a.n_conds : 0, dmax, sizeof(T)),
                                                   l16, K, dmax, gloss);
    }

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
// small k (fp32 state, k < 64: the reference's default num.factors is 10): LPT = 4, 8 or 16 lanes per unit
// ---------------------------------------------------------------------------------------------
// Same walk as sgd_chain_level with the lane layout of sgd_level_small_f32 (mf_sgd_kernels.hip): lane l of the group owns
// factors l, l+LPT, l+2*LPT, l+3*LPT (scalar loads: rows of k floats are not 16-byte aligned for general k) and the tuple's l-th
// condition, so LPT >= dmax and 4*LPT >= k; 256/LPT units per workgroup.  Per tuple the expressions are small_tuples_f32's, in
// the same order (bit-identical fp32 model).  At this k a level is a few megabytes, i.e. the epoch is bound by the number of
// dependent launches -- which is what the chain schedule cuts (939 -> 285 for the C3 shape).
struct SmallSpoke {
    float v[4];
    float sb, sc;
    float *psc;
    int spoke, cond;
};

template <int MODEL, int LPT, bool HUB_ITEM>
__device__ __forceinline__ void small_load_spoke(const SgdArgs<float> &a, int spoke, int cond, int lt, int k, SmallSpoke &r) {
    using M = Traits<MODEL>;
    constexpr bool SB = HUB_ITEM ? M::has_bu : M::has_bj;
    constexpr bool SC = HUB_ITEM ? M::has_uc : M::has_ic;
    const float *row = (HUB_ITEM ? a.P : a.Q) + (size_t)spoke * k + lt;
    r.spoke = spoke;
    r.cond = cond;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        r.v[v] = 0.f;
        if (lt + v * LPT < k) r.v[v] = row[v * LPT];
    }
    r.sb = r.sc = 0.f;
    r.psc = nullptr;
    if (SB) r.sb = (HUB_ITEM ? a.userBias : a.itemBias)[spoke];
    if (SC && cond >= 0) {
        r.psc = (HUB_ITEM ? a.ucBias : a.icBias) + (size_t)spoke * a.n_conds + cond;
        r.sc = *r.psc;
    }
}

template <int MODEL, int LPT, bool HUB_ITEM>
__device__ __forceinline__ void small_step(const SgdArgs<float> &a, const ChainHp<float> &hp, float (&h)[4], float &hb, float *s_hc,
                                           const SmallSpoke &cur, float rr, int lt, int k, double &gloss) {
    using M = Traits<MODEL>;
    constexpr bool HB = HUB_ITEM ? M::has_bj : M::has_bu;
    constexpr bool SB = HUB_ITEM ? M::has_bu : M::has_bj;
    constexpr bool HC = HUB_ITEM ? M::has_ic : M::has_uc;
    constexpr bool SC = HUB_ITEM ? M::has_uc : M::has_ic;
    const float lr = hp.lr, regU = hp.regU, regI = hp.regI, regB = hp.regB, regC = hp.regC;
    float part = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) part += HUB_ITEM ? cur.v[v] * h[v] : h[v] * cur.v[v];
    const float dot = group_sum<LPT>(part);
    float hcv = 0.f;
    if (HC) {
        chain_lds_order();
        if (cur.cond >= 0) hcv = s_hc[cur.cond];
    }
    const float bu = HUB_ITEM ? cur.sb : hb, bj = HUB_ITEM ? hb : cur.sb;
    const float bic = HUB_ITEM ? hcv : cur.sc, buc = HUB_ITEM ? cur.sc : hcv;
    float pred = hp.gm;
    if (M::has_bu) pred += bu;
    if (M::has_bj) pred += bj;
    pred += dot;
    if (M::has_ctx) {
        float term = 0.f;
        if (M::has_ic && M::has_uc) term = bic + buc;
        else if (M::has_ic) term = bic;
        else if (M::has_uc) term = buc;
        pred += group_sum<LPT>(term);
    }
    const float e = rr - pred;
    if (HB) hb = hb + lr * (e - regB * hb);
    if (SB && lt == 0) (HUB_ITEM ? a.userBias : a.itemBias)[cur.spoke] = cur.sb + lr * (e - regB * cur.sb);
    float ctx_loss = 0.f;
    if (cur.cond >= 0) {
        if (M::has_ic) ctx_loss += bic * bic;
        if (M::has_uc) ctx_loss += buc * buc;
        if (HC) s_hc[cur.cond] = hcv + lr * (e - regC * hcv);
        if (SC) *cur.psc = cur.sc + lr * (e - regC * cur.sc);
    }
    if (HC) chain_lds_order();
    float lsum = 0.f;
    float *srow = (HUB_ITEM ? a.P : a.Q) + (size_t)cur.spoke * k + lt;
#pragma unroll
    for (int v = 0; v < 4; ++v)
        if (lt + v * LPT < k) {
            const float pv = HUB_ITEM ? cur.v[v] : h[v], qv = HUB_ITEM ? h[v] : cur.v[v];
            const float pn = pv + lr * (e * qv - regU * pv);
            const float qn = qv + lr * (e * pv - regI * qv);
            lsum += (regU * pv) * pv + (regI * qv) * qv;
            h[v] = HUB_ITEM ? qn : pn;
            srow[v * LPT] = HUB_ITEM ? pn : qn;
        }
    const float reg_loss = group_sum<LPT>(lsum);
    const float ctx_sum = M::has_ctx ? group_sum<LPT>(ctx_loss) : 0.f;
    if (lt == 0) {
        double l = (double)e * (double)e;
        if (M::has_bu) l += (double)regB * bu * bu;
        if (M::has_bj) l += (double)regB * bj * bj;
        if (M::has_ctx) l += (double)regC * ctx_sum;
        gloss += l + (double)reg_loss;
    }
}

template <int MODEL, int LPT, bool HUB_ITEM>
__global__ __launch_bounds__(256) void sgd_chain_small(SgdArgs<float> a, const int32_t *__restrict__ unit_off, int64_t ubegin, int count,
                                                       int64_t slot0) {
    using M = Traits<MODEL>;
    static_assert(MODEL != CAMF_C, "CAMF_C has no level schedule (shared condBias)");
    constexpr bool HB = HUB_ITEM ? M::has_bj : M::has_bu;
    constexpr bool HC = HUB_ITEM ? M::has_ic : M::has_uc;
    constexpr int GPB = 256 / LPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
    __shared__ double s_loss[GPB];
    const int tid = threadIdx.x, lt = tid % LPT, gib = tid / LPT;
    const int g = blockIdx.x * GPB + gib;
    const int k = a.k;
    const int dmax = M::has_ctx ? a.dmax : 0;
    double gloss = 0.0;
    const HParams hpd = *a.hp;
    const ChainHp<float> hp{(float)hpd.lr, (float)hpd.regU, (float)hpd.regI, (float)hpd.regB, (float)hpd.regC, (float)hpd.gm};

    if (g < count) { // group-uniform
        const int32_t tb = unit_off[ubegin + g], te = unit_off[ubegin + g + 1];
        const int len = te - tb; // 1 .. 16
        unsigned char *gbase = chain_smem + (size_t)gib * chain_group_lds(HC ? a.n_conds : 0, dmax, sizeof(float));
        float *s_hc = reinterpret_cast<float *>(gbase);
        float *s_rr = s_hc + (HC ? a.n_conds : 0);
        int *s_sp = reinterpret_cast<int *>(s_rr + 16);
        int *s_cd = s_sp + 16;
        const int n_cd = len * dmax;

        // ---- ids: every load issued before the first is used (16 / LPT passes over the unit's tuples, <= 16 over its conditions)
        const int hub = HUB_ITEM ? a.sj[tb] : a.su[tb];
        const int sp0 = HUB_ITEM ? a.su[tb] : a.sj[tb];
        const int cd0 = lt < dmax ? a.sconds[(int64_t)tb * dmax + lt] : -1;
        int my_sp[16 / LPT];
        float my_rr[16 / LPT];
#pragma unroll
        for (int r = 0; r < 16 / LPT; ++r) {
            my_sp[r] = 0;
            my_rr[r] = 0.f;
            if (lt + r * LPT < len) {
                my_sp[r] = HUB_ITEM ? a.su[tb + lt + r * LPT] : a.sj[tb + lt + r * LPT];
                my_rr[r] = a.sr[tb + lt + r * LPT];
            }
        }
        int cdv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cdv[r] = -1;
            if (lt + r * LPT < n_cd) cdv[r] = a.sconds[(int64_t)tb * dmax + lt + r * LPT];
        }
        // ---- the hub row comes on chip once, together with the first spoke row
        float *hrow = (HUB_ITEM ? a.Q : a.P) + (size_t)hub * k + lt;
        float h[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            h[v] = 0.f;
            if (lt + v * LPT < k) h[v] = hrow[v * LPT];
        }
        float hb = 0.f;
        float *phb = nullptr;
        if (HB) {
            phb = (HUB_ITEM ? a.itemBias : a.userBias) + hub;
            hb = *phb;
        }
        float *hc_row = HC ? (HUB_ITEM ? a.icBias : a.ucBias) + (size_t)hub * a.n_conds : nullptr;
        SmallSpoke A, B;
        small_load_spoke<MODEL, LPT, HUB_ITEM>(a, sp0, cd0, lt, k, A);
        // ---- ids and the hub's context-bias row into LDS
#pragma unroll
        for (int r = 0; r < 16 / LPT; ++r)
            if (lt + r * LPT < len) {
                s_sp[lt + r * LPT] = my_sp[r];
                s_rr[lt + r * LPT] = my_rr[r];
            }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (lt + r * LPT < n_cd) s_cd[lt + r * LPT] = cdv[r];
        for (int c = lt + 4 * LPT; c < n_cd; c += LPT) s_cd[c] = a.sconds[(int64_t)tb * dmax + c];
        if (HC)
            for (int c = lt; c < a.n_conds; c += LPT) s_hc[c] = hc_row[c];
        chain_lds_order();

        int i = 0;
        while (true) {
            if (i + 1 < len) small_load_spoke<MODEL, LPT, HUB_ITEM>(a, s_sp[i + 1], lt < dmax ? s_cd[(i + 1) * dmax + lt] : -1, lt, k, B);
            small_step<MODEL, LPT, HUB_ITEM>(a, hp, h, hb, s_hc, A, s_rr[i], lt, k, gloss);
            if (++i >= len) break;
            if (i + 1 < len) small_load_spoke<MODEL, LPT, HUB_ITEM>(a, s_sp[i + 1], lt < dmax ? s_cd[(i + 1) * dmax + lt] : -1, lt, k, A);
            small_step<MODEL, LPT, HUB_ITEM>(a, hp, h, hb, s_hc, B, s_rr[i], lt, k, gloss);
            if (++i >= len) break;
        }
        // ---- the hub row leaves the chip once
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (lt + v * LPT < k) hrow[v * LPT] = h[v];
        if (HB && lt == 0) *phb = hb;
        if (HC) {
            chain_lds_order();
            for (int c = lt; c < a.n_conds; c += LPT) hc_row[c] = s_hc[c];
        }
    }
    if (lt == 0) s_loss[gib] = gloss;
    __syncthreads();
    if (tid < 64) { // fixed-shape tree over the GPB group sums
        double sum = 0.0;
        for (int q = tid; q < GPB; q += 64) sum += s_loss[q];
        sum = wave_sum64(sum);
        if (tid == 0) a.loss_part[slot0 + blockIdx.x] = sum;
    }
}

// ---------------------------------------------------------------------------------------------
// spoke arena <-> model table: one 16-byte vector per thread, a row's first tuple slot holds its live value between epochs
// ---------------------------------------------------------------------------------------------
template <typename T, bool TO_ARENA>
__global__ __launch_bounds__(256) void arena_rows_kernel(T *table, T *arena, const int32_t *__restrict__ first_pos, int64_t n_rows, int k) {
    using V = typename Vec16<T>::type;
    constexpr int E = Vec16<T>::E;
    const int vec_per_row = k / E;                       // k % E == 0 (the chain kernels' own requirement for vector rows)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = i / vec_per_row;
    const int v = (int)(i % vec_per_row);
    if (row >= n_rows) return;
    const int32_t p = first_pos[row];
    if (p < 0) return;                                   // a row without tuples never enters the arena
    V *t = reinterpret_cast<V *>(table + (size_t)row * k) + v;
    V *a = reinterpret_cast<V *>(arena + (size_t)p * k) + v;
    if (TO_ARENA) *a = *t;
    else *t = *a;
}
template <typename T>
hipError_t launch_arena_scatter(const T *table, T *arena, const int32_t *first_pos, int64_t n_rows, int k, hipStream_t s) {
    const int64_t n = n_rows * (k / Vec16<T>::E);
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL((arena_rows_kernel<T, true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, const_cast<T *>(table), arena, first_pos, n_rows, k);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_arena_gather(T *table, const T *arena, const int32_t *first_pos, int64_t n_rows, int k, hipStream_t s) {
    const int64_t n = n_rows * (k / Vec16<T>::E);
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL((arena_rows_kernel<T, false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, table, const_cast<T *>(arena), first_pos, n_rows, k);
    return hipGetLastError();
}
template hipError_t launch_arena_scatter<float>(const float *, float *, const int32_t *, int64_t, int, hipStream_t);
template hipError_t launch_arena_scatter<double>(const double *, double *, const int32_t *, int64_t, int, hipStream_t);
template hipError_t launch_arena_gather<float>(float *, const float *, const int32_t *, int64_t, int, hipStream_t);
template hipError_t launch_arena_gather<double>(double *, const double *, const int32_t *, int64_t, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------

bool has_chain_path(int model, int k, int dmax, int n_conds, bool f64, bool strict) {
    if (strict || model == CAMF_C) return false;
    if (dmax > 16) return false;
    if (f64) {
        if (k < 32 || k > 256 || k % 2 != 0) return false;
    } else {
        if (k > 256 || (k >= 64 && k % 4 != 0)) return false; // k < 64: sgd_chain_small (any k)
    }
    const int groups = chain_groups_per_block(k, dmax, f64);
    if (groups * chain_lds_bytes(model, n_conds, dmax, f64, true) / 16 > 64 * 1024 || groups * chain_lds_bytes(model, n_conds, dmax, f64, false) / 16 > 64 * 1024)
        return false;
    return true;
}

size_t chain_lds_bytes(int model, int n_conds, int dmax, bool f64, bool hub_is_item) {
    const bool has_ic = model == CAMF_CI || model == CAMF_CUCI, has_uc = model == CAMF_CU || model == CAMF_CUCI;
    const bool hc = hub_is_item ? has_ic : has_uc;
    const bool has_ctx = model != BIASEDMF && model != PMF;
    return 16 * chain_group_lds(hc ? n_conds : 0, has_ctx ? dmax : 0, f64 ? 8 : 4);
}

// lanes per unit of the small-k kernel (as small_lpt in mf_sgd_kernels.hip): LPT >= dmax and 4 * LPT >= k
static int chain_small_lpt(int k, int dmax) {
    if (k <= 16 && dmax <= 4) return 4;
    if (k <= 32 && dmax <= 8) return 8;
    return 16;
}
int chain_groups_per_block(int k, int dmax, bool f64) { return (!f64 && k < 64) ? 256 / chain_small_lpt(k, dmax) : 16; }
int chain_level_blocks(int k, int dmax, bool f64, int count) {
    const int g = chain_groups_per_block(k, dmax, f64);
    return (count + g - 1) / g;
}

template <typename T, int MODEL, int NV, bool RAGGED>
static void *chain_kernel_hub(bool hub_is_item) {
    return hub_is_item ? (void *)sgd_chain_level<T, MODEL, NV, RAGGED, true> : (void *)sgd_chain_level<T, MODEL, NV, RAGGED, false>;
}

template <typename T, int MODEL>
static void *chain_kernel_k(int k, bool hub_is_item) {
    constexpr int E = Vec16<T>::E;
    const int per = 16 * E; // factors one vector slot covers across the group
    if (k % per == 0) {
        switch (k / per) {
        case 1: return chain_kernel_hub<T, MODEL, 1, false>(hub_is_item);
        case 2: return chain_kernel_hub<T, MODEL, 2, false>(hub_is_item);
        case 4: return chain_kernel_hub<T, MODEL, 4, false>(hub_is_item);
        case 8: if (E == 2) return chain_kernel_hub<T, MODEL, 8, false>(hub_is_item); break;
        }
    }
    const int nv = (k + per - 1) / per; // masked vector slots past k
    if (nv <= 2) return chain_kernel_hub<T, MODEL, 2, true>(hub_is_item);
    if (nv <= 3) return chain_kernel_hub<T, MODEL, 3, true>(hub_is_item);
    if (nv <= 4) return chain_kernel_hub<T, MODEL, 4, true>(hub_is_item);
    if (E == 2) {
        if (nv <= 6) return chain_kernel_hub<T, MODEL, 6, true>(hub_is_item);
        if (nv <= 8) return chain_kernel_hub<T, MODEL, 8, true>(hub_is_item);
    }
    return nullptr;
}

template <typename T>
static void *chain_kernel_ptr(int model, int k, bool hub_is_item) {
    switch (model) {
    case BIASEDMF: return chain_kernel_k<T, BIASEDMF>(k, hub_is_item);
    case PMF: return chain_kernel_k<T, PMF>(k, hub_is_item);
    case CAMF_CI: return chain_kernel_k<T, CAMF_CI>(k, hub_is_item);
    case CAMF_CU: return chain_kernel_k<T, CAMF_CU>(k, hub_is_item);
    case CAMF_CUCI: return chain_kernel_k<T, CAMF_CUCI>(k, hub_is_item);
    }
    return nullptr;
}

template <int MODEL, int LPT>
static void *chain_small_hub(bool hub_is_item) {
    return hub_is_item ? (void *)sgd_chain_small<MODEL, LPT, true> : (void *)sgd_chain_small<MODEL, LPT, false>;
}
template <int MODEL>
static void *chain_small_lpt_ptr(int lpt, bool hub_is_item) {
    switch (lpt) {
    case 4: return chain_small_hub<MODEL, 4>(hub_is_item);
    case 8: return chain_small_hub<MODEL, 8>(hub_is_item);
    default: return chain_small_hub<MODEL, 16>(hub_is_item);
    }
}
static void *chain_small_ptr(int model, int lpt, bool hub_is_item) {
    switch (model) {
    case BIASEDMF: return chain_small_lpt_ptr<BIASEDMF>(lpt, hub_is_item);
    case PMF: return chain_small_lpt_ptr<PMF>(lpt, hub_is_item);
    case CAMF_CI: return chain_small_lpt_ptr<CAMF_CI>(lpt, hub_is_item);
    case CAMF_CU: return chain_small_lpt_ptr<CAMF_CU>(lpt, hub_is_item);
    case CAMF_CUCI: return chain_small_lpt_ptr<CAMF_CUCI>(lpt, hub_is_item);
    }
    return nullptr;
}

template <typename T>
hipError_t launch_chain_level(const SgdArgs<T> &a, const LaunchCfg &cfg, bool hub_is_item, const int32_t *unit_off, int64_t ubegin,
                              int count, int64_t slot0, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    const bool f64 = sizeof(T) == 8;
    const int groups = chain_groups_per_block(a.k, a.dmax, f64);
    void *fn = (!f64 && a.k < 64) ? chain_small_ptr(cfg.model, 256 / groups, hub_is_item) : chain_kernel_ptr<T>(cfg.model, a.k, hub_is_item);
    if (!fn) return hipErrorInvalidValue;
    SgdArgs<T> args = a;
    void *params[] = {&args, &unit_off, &ubegin, &count, &slot0};
    const size_t lds = (size_t)groups * (chain_lds_bytes(cfg.model, a.n_conds, a.dmax, f64, hub_is_item) / 16);
    return hipLaunchKernel(fn, dim3((unsigned)chain_level_blocks(a.k, a.dmax, f64, count)), dim3(256), params, lds, s);
}
template hipError_t launch_chain_level<float>(const SgdArgs<float> &, const LaunchCfg &, bool, const int32_t *, int64_t, int, int64_t,
                                              hipStream_t);
template hipError_t launch_chain_level<double>(const SgdArgs<double> &, const LaunchCfg &, bool, const int32_t *, int64_t, int, int64_t,
                                               hipStream_t);

} // namespace cmi

// owner_kernels.hip -- the owner (dataflow) epoch for heavy-tailed degrees: ONE persistent launch, no levels.
//
// Schedule: level_schedule.cpp, build_owner_schedule.  Every hub row (an item, or a user) has one owner for the whole epoch: a
// wavefront of sgd_owner.  The owner walks the tuples of all its rows in CRS order (SGD order of the reference, IterativeRecommender /
// CAMF_CI.java:66-121 and siblings), which makes the epoch order-exact by construction:
//   * hub side (Q row, itemBias, icBias row when items are owned): private to the owner, plain loads/stores; while consecutive tuples
//     of the list share the hub row it stays in registers -- the chain along the hottest row, which a level schedule pays a launch
//     or a barrier per link for, costs one dot product + one axpy of latency per link here;
//   * spoke side (P row, userBias, ucBias row): the row's RECORD in a tagged copy of the table.  A record is a run of 8-byte granules
//     {32 data bits, tag}, written by device-coherent (sc0 sc1, write-through) stores and read by L1-bypassing loads, one or two
//     whole aligned granules per lane and instruction; the tag is the number of updates applied to the row so far this epoch.  The tuple that needs update count `want` may use the record
//     once EVERY granule carries tag == want: the data is the flag, so there is no separate version word, no store drain and no
//     fence (CDNA4 hand-off recipe R2: data-tagged granules).  A granule is never torn (one aligned 8-byte store) and a record in
//     mid-update simply fails the test.  The updated record is written back with tag want + 1.
//   * records are read D tuples AHEAD of their use (speculatively: a record that was not ready yet fails its tag test at use, and
//     only then does the owner poll) -- the owner of a hot row is bound by the arithmetic chain, not by HBM latency.
// A tag pass before the launch builds the records (tag 0) from the model tables and an untag pass after it writes them back, both
// at copy speed.  Everything in sgd_owner is wave-uniform: one tuple per wavefront step, the tuple's fields arrive through scalar
// loads, and lane l holds elements l VPL ... l VPL + VPL - 1 of each row, conditions l, l + 64, ... of the context-bias rows (NCW words).
// Two forms share the launch: four one-wavefront owners per workgroup (sgd_owner's body), and, for the owners of the hottest rows, a
// workgroup per owner whose three wavefronts split the step (owner_team: loader -> LDS ring -> compute -> LDS ring -> storer).
// With CMI_FLAG_STRICT (fp64) the step uses the reference's operation order throughout and the model is bit-identical to the oracle's.
#include "sgd_device.hpp"
#include "env_knobs.hpp"
#include "level_schedule.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

namespace cmi {

typedef unsigned long long gran_t;

__device__ __forceinline__ gran_t ld_gran(const gran_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_gran(gran_t *p, gran_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One model element (T) as tagged granules: fp32 = one granule, fp64 = two (low word, high word).
template <typename T>
struct Tagged;
template <>
struct Tagged<float> {
    static constexpr int NW = 1;
    struct Raw {
        gran_t g0;
    };
    static __device__ __forceinline__ Raw zero() { return Raw{0ull}; }
    static __device__ __forceinline__ Raw load(const gran_t *rec, int e) { return Raw{ld_gran(rec + e)}; }
    static __device__ __forceinline__ bool ok(const Raw &r, uint32_t want) { return (uint32_t)(r.g0 >> 32) == want; }
    static __device__ __forceinline__ float value(const Raw &r) { return __uint_as_float((uint32_t)r.g0); }
    static __device__ __forceinline__ void store(gran_t *rec, int e, float v, uint32_t tag) {
        st_gran(rec + e, ((gran_t)tag << 32) | (gran_t)__float_as_uint(v));
    }
    static __device__ __forceinline__ void store_plain(gran_t *rec, int e, float v, uint32_t tag) {
        rec[e] = ((gran_t)tag << 32) | (gran_t)__float_as_uint(v);
    }
};
template <>
struct Tagged<double> {
    static constexpr int NW = 2;
    struct Raw {
        gran_t g0, g1;
    };
    static __device__ __forceinline__ Raw zero() { return Raw{0ull, 0ull}; }
    static __device__ __forceinline__ Raw load(const gran_t *rec, int e) { return Raw{ld_gran(rec + 2 * e), ld_gran(rec + 2 * e + 1)}; }
    static __device__ __forceinline__ bool ok(const Raw &r, uint32_t want) {
        return (uint32_t)(r.g0 >> 32) == want && (uint32_t)(r.g1 >> 32) == want;
    }
    static __device__ __forceinline__ double value(const Raw &r) {
        return __longlong_as_double((long long)(((gran_t)(uint32_t)r.g1 << 32) | (gran_t)(uint32_t)r.g0));
    }
    static __device__ __forceinline__ void store(gran_t *rec, int e, double v, uint32_t tag) {
        const gran_t b = (gran_t)__double_as_longlong(v);
        st_gran(rec + 2 * e, ((gran_t)tag << 32) | (b & 0xffffffffull));
        st_gran(rec + 2 * e + 1, ((gran_t)tag << 32) | (b >> 32));
    }
    static __device__ __forceinline__ void store_plain(gran_t *rec, int e, double v, uint32_t tag) {
        const gran_t b = (gran_t)__double_as_longlong(v);
        rec[2 * e] = ((gran_t)tag << 32) | (b & 0xffffffffull);
        rec[2 * e + 1] = ((gran_t)tag << 32) | (b >> 32);
    }
};

// Which containers sit on which side (spoke = the side that travels in records).
template <int MODEL, bool HUB_ITEM>
struct Sides {
    using M = Traits<MODEL>;
    static constexpr bool HB = HUB_ITEM ? M::has_bj : M::has_bu; // hub scalar bias
    static constexpr bool SB = HUB_ITEM ? M::has_bu : M::has_bj; // spoke scalar bias
    static constexpr bool HC = HUB_ITEM ? M::has_ic : M::has_uc; // hub context-bias row
    static constexpr bool SC = HUB_ITEM ? M::has_uc : M::has_ic; // spoke context-bias row
};

// Record layout, in elements: [0, 64 VPL) the factor row PADDED to whole wavefronts (lane l owns elements l VPL .. l VPL + VPL - 1),
// then 64 NCW elements of the context-bias row (lane l holds conditions l, l + 64, ...) when the spoke side has one, then the scalar bias.  The padding is
// what lets the steady state run without a single masked access: every lane loads and stores all its granules, the elements past k
// (past n_conds) are zeros that stay zero under the update.
__host__ __device__ inline int64_t owner_record_granules(int vpl, int ncw, bool sb, int nw) { // ncw: 64-condition words of the context row (0: none)
    const int64_t g = (int64_t)(64 * vpl + 64 * ncw + (sb ? 1 : 0)) * nw;
    return (g + 15) & ~(int64_t)15; // 128-byte multiples
}
__host__ __device__ inline int owner_vpl(int k) { return k <= 64 ? 1 : (k <= 128 ? 2 : 4); }

// Sum over the wavefront as a uniform value: DPP row rotations (every lane of a 16-lane row gets the row total), then row 0 into
// row 1 and row 2 into row 3 (row_bcast:15), rows 0+1 into row 3 (row_bcast:31); lane 63 holds ((r2 + r3) + (r0 + r1)).
__device__ __forceinline__ float wave_total(float x) {
    x = row_sum16(x);
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x142, 0xa, 0xf, false));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x143, 0xc, 0xf, false));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ double wave_total(double x) {
    x = row_sum16(x);
    x += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x142, 0xa, 0xf, false),
                          __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x142, 0xa, 0xf, false));
    x += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x143, 0xc, 0xf, false),
                          __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x143, 0xc, 0xf, false));
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}

// ---------------------------------------------------------------------------------------------
// tag / untag passes: model tables <-> tagged records (one wavefront per record, grid-stride)
// ---------------------------------------------------------------------------------------------
template <typename T, bool TO_RECORDS>
__global__ __launch_bounds__(256) void owner_records(T *__restrict__ rows, T *__restrict__ ctx, T *__restrict__ bias, gran_t *__restrict__ tagged,
                                                     int64_t stride, int n_spokes, int k, int ncs, int vpl, int ncw, uint32_t tag0) {
    const int lane = threadIdx.x & 63;
    const int row_cap = 64 * vpl;
    const int64_t waves = (int64_t)gridDim.x * 4;
    for (int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); s < n_spokes; s += waves) {
        gran_t *rec = tagged + s * stride;
        for (int e = lane; e < row_cap; e += 64) {
            if (TO_RECORDS) Tagged<T>::store_plain(rec, e, e < k ? rows[s * k + e] : (T)0, tag0);
            else if (e < k) rows[s * k + e] = Tagged<T>::value(Tagged<T>::load(rec, e));
        }
        if (ctx)
            for (int c = lane; c < 64 * ncw; c += 64) {
                if (TO_RECORDS) Tagged<T>::store_plain(rec, row_cap + c, c < ncs ? ctx[s * ncs + c] : (T)0, tag0);
                else if (c < ncs) ctx[s * ncs + c] = Tagged<T>::value(Tagged<T>::load(rec, row_cap + c));
            }
        if (bias && lane == 0) {
            const int e = row_cap + (ctx ? 64 * ncw : 0);
            if (TO_RECORDS) Tagged<T>::store_plain(rec, e, bias[s], tag0);
            else bias[s] = Tagged<T>::value(Tagged<T>::load(rec, e));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// the epoch
// ---------------------------------------------------------------------------------------------
#define CMI_OWNER_SPIN_LIMIT (1u << 24)
// One more poll of a wait loop.  True = leave the loop: either this wait has exhausted its bound (the epoch is flagged as stalled:
// error[0] = 1) or ANOTHER owner has already flagged it -- then the model state is lost anyway and every later wait gives up after
// its first poll instead of spinning to the bound again.  The flag is read once per 4096 polls: the cold path of a cold path.
__device__ __forceinline__ bool owner_spin_expired(unsigned &spins, int *error, int lane) {
    if ((spins++ & 4095u) == 0 && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
    if (spins > CMI_OWNER_SPIN_LIMIT) {
        if (lane == 0) atomicExch(error, 1);
        return true;
    }
    return false;
}
#ifdef CMI_OWNER_TRACE
// Debug builds only (make TRACE=1): every owner writes what it read and what it produced for every tuple into SgdArgs::trace -- per list
// position CMI_TR_ROWS rows of 64 doubles (lane l = element l): 0 hub row in, 1 spoke row in, 2 hub context row in, 3 {spoke bias in, off,
// hub, want, flags, 1, tag0}, 4-7 the same after the update (7: {spoke bias out}), 8 / 9 the team loader's view of the spoke row / bias,
// 10 / 11 the team storer's.  tools/exp/owner_trace.py compares the trace of a lone run with that of a run beside other owner epochs.
#define CMI_TR_ROWS 12
template <typename T>
__device__ __forceinline__ void tr_put(double *tr, int64_t pos, int row, int lane, T v) {
    if (tr) tr[((size_t)pos * CMI_TR_ROWS + row) * 64 + lane] = (double)v;
}
#define CMI_TR(...) __VA_ARGS__
#else
#define CMI_TR(...)
#endif
static const int OWNER_DEPTH_MAX = 16; // every list is followed by OWNER_DEPTH_MAX + 1 inert entries (cmi_api.cpp), whatever D a kernel uses

// Spoke records go through buffer instructions: one resource over the record table, the record's byte offset in the scalar offset
// (no address arithmetic per step), 16 bytes = two granules per lane and instruction, device-coherent (sc0 sc1: write-through
// stores, L1-bypassing loads).  A 16-byte access is two aligned granules; each granule stays whole.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CMI_OWNER_CPOL 17 /* sc0 sc1 */

// W words (W = 2: one granule, 4: two, 8: four) of this lane at byte `voff` of the record at byte `soff` of the table
template <int W>
__device__ __forceinline__ void owner_ld_words(__amdgpu_buffer_rsrc_t rs, int voff, int soff, uint32_t (&w)[W]) {
    static_assert(W == 2 || W == 4 || W == 8, "one, two or four granules");
    if constexpr (W == 2) {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, CMI_OWNER_CPOL);
        w[0] = t.x, w[1] = t.y;
    } else {
#pragma unroll
        for (int i = 0; i < W / 4; ++i) {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16 * i, soff, CMI_OWNER_CPOL);
            w[4 * i] = t.x, w[4 * i + 1] = t.y, w[4 * i + 2] = t.z, w[4 * i + 3] = t.w;
        }
    }
}
template <int W>
__device__ __forceinline__ void owner_st_words(__amdgpu_buffer_rsrc_t rs, int voff, int soff, const uint32_t (&w)[W]) {
    if constexpr (W == 2) {
        u32x2 t;
        t.x = w[0], t.y = w[1];
        __builtin_amdgcn_raw_buffer_store_b64(t, rs, voff, soff, CMI_OWNER_CPOL);
    } else {
#pragma unroll
        for (int i = 0; i < W / 4; ++i) {
            u32x4 t;
            t.x = w[4 * i], t.y = w[4 * i + 1], t.z = w[4 * i + 2], t.w = w[4 * i + 3];
#if defined(CMI_VAR_STORE_B64)
            u32x2 lo, hi; // experiment: the same 16 bytes as two 8-byte stores (no > 64-bit store data)
            lo.x = t.x, lo.y = t.y, hi.x = t.z, hi.y = t.w;
            __builtin_amdgcn_raw_buffer_store_b64(lo, rs, voff + 16 * i, soff, CMI_OWNER_CPOL);
            __builtin_amdgcn_raw_buffer_store_b64(hi, rs, voff + 16 * i + 8, soff, CMI_OWNER_CPOL);
#else
            __builtin_amdgcn_raw_buffer_store_b128(t, rs, voff + 16 * i, soff, CMI_OWNER_CPOL);
#if !defined(CMI_VAR_NO_STORE_NOP)
            // STORE-DATA HAZARD (gfx9 family, "VMEM store of more than 64 bits followed by a VALU write of the VGPRs that hold its data"):
            // the store reads its data registers AFTER it has issued, so a v_mov into one of them in the next one or two issue slots can
            // reach the register first and the NEW value is stored.  The compiler's hazard recognizer inserts the wait states itself --
            // except for buffer stores whose soffset is an SGPR (it takes the ISA manual's word that those are exempt), which is exactly
            // this store (the record's byte offset travels in soffset).  Measured on gfx950 they are not exempt: with the data registers
            // reused at once (the bias granules are packed into the row's registers: `buffer_store_dwordx4 v[2:5] ... ; v_mov_b32 v2, v74`)
            // a record now and then went out with the right tags and the NEXT store's low word in 16 lanes -- only when another
            // workgroup's memory instructions delayed the read, i.e. beside another owner epoch (docs/history/r06.md 1;
            // tools/micro/store_data_hazard.hip reproduces it in isolation).  The asm below reads the store's data TUPLE (the same four registers: as four
            // scalar operands the compiler found the values elsewhere and still rewrote the tuple before the asm), so they stay live --
            // unwritten -- up to it, and s_nop 1 supplies the two wait states the manual asks for on gfx940+.
            asm volatile("s_nop 1" : : "v"(t) : "memory");
#endif
#endif
        }
    }
}
// element i of a word array {data, tag, data, tag, ...} (fp64: {low, tag, high, tag})
__device__ __forceinline__ float owner_elem(const uint32_t *w, int i, float) { return __uint_as_float(w[2 * i]); }
__device__ __forceinline__ double owner_elem(const uint32_t *w, int i, double) {
    return __hiloint2double((int)w[4 * i + 2], (int)w[4 * i]);
}
__device__ __forceinline__ void owner_pack(uint32_t *w, int i, float v, uint32_t tag) { w[2 * i] = __float_as_uint(v), w[2 * i + 1] = tag; }
__device__ __forceinline__ void owner_pack(uint32_t *w, int i, double v, uint32_t tag) {
    w[4 * i] = (uint32_t)__double2loint(v), w[4 * i + 1] = tag, w[4 * i + 2] = (uint32_t)__double2hiint(v), w[4 * i + 3] = tag;
}

// what a list position prefetches, D positions ahead of its use
template <typename T, int VPL, int NCW>
struct OwnerSlot {
    static constexpr int NW = Tagged<T>::NW;
    uint32_t xw[VPL * NW * 2], cw[NCW][NW * 2], bw[NW * 2]; // spoke record: this lane's row elements, context biases of conditions lane + 64 w, bias
    T hq[VPL], hc[NCW], hb;                                 // hub side, plain
};

template <typename T, int MODEL, int VPL, int NCW, bool HUB_ITEM>
__device__ __forceinline__ void owner_load_spoke(__amdgpu_buffer_rsrc_t rs, int soff, int lane, OwnerSlot<T, VPL, NCW> &s) {
    using S = Sides<MODEL, HUB_ITEM>;
    constexpr int NW = Tagged<T>::NW;
    owner_ld_words(rs, lane * (VPL * NW * 8), soff, s.xw);
    if (S::SC) {
#pragma unroll
        for (int w = 0; w < NCW; ++w) owner_ld_words(rs, (64 * VPL + 64 * w) * NW * 8 + lane * (NW * 8), soff, s.cw[w]);
    }
    if (S::SB) owner_ld_words(rs, (64 * VPL + (S::SC ? 64 * NCW : 0)) * NW * 8, soff, s.bw); // the same granule(s) in every lane
}

template <typename T, int MODEL, int VPL, int NCW, bool HUB_ITEM>
__device__ __forceinline__ bool owner_spoke_ok(const OwnerSlot<T, VPL, NCW> &s, uint32_t want) {
    using S = Sides<MODEL, HUB_ITEM>;
    constexpr int NW = Tagged<T>::NW;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < VPL * NW; ++i) ok &= s.xw[2 * i + 1] == want;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        if (S::SC) {
#pragma unroll
            for (int w = 0; w < NCW; ++w) ok &= s.cw[w][2 * i + 1] == want;
        }
        if (S::SB) ok &= s.bw[2 * i + 1] == want;
    }
    return ok;
}

// hub side: the model tables themselves.  Loads are unmasked (a lane past the end of the row reads the row's last element instead and
// the value is zeroed where it is taken into the working registers): a masked load would need a zeroed default, and the copy that
// merges the two makes the compiler wait for the load on the spot.
template <typename T, int MODEL, int VPL, int NCW, bool HUB_ITEM>
__device__ __forceinline__ void owner_load_hub(const SgdArgs<T> &a, int hub, int lane, int k, T (&hq)[VPL], T (&hc)[NCW], T &hb) {
    using S = Sides<MODEL, HUB_ITEM>;
    const T *row = (HUB_ITEM ? a.Q : a.P) + (size_t)hub * k;
#pragma unroll
    for (int v = 0; v < VPL; ++v) hq[v] = row[min(lane * VPL + v, k - 1)];
    if (S::HC) {
#pragma unroll
        for (int w = 0; w < NCW; ++w) hc[w] = (HUB_ITEM ? a.icBias : a.ucBias)[(size_t)hub * a.n_conds + min(lane + 64 * w, a.n_conds - 1)];
    }
    if (S::HB) hb = (HUB_ITEM ? a.itemBias : a.userBias)[hub];
}

__device__ __forceinline__ float owner_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double owner_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// make the compiler finish the loads of `v` here (a wait it places before an instruction that reads the register)
__device__ __forceinline__ void owner_settle(float &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void owner_settle(double &v) { asm volatile("" : "+v"(v)); }

template <typename T>
struct OwnerRating;
template <>
struct OwnerRating<float> {
    template <int NCW>
    static __device__ __forceinline__ float get(const OwnerRecT<NCW> &r) { return r.rating.f; }
};
template <>
struct OwnerRating<double> {
    template <int NCW>
    static __device__ __forceinline__ double get(const OwnerRecT<NCW> &r) { return r.rating.d; }
};

// value of lane l (wave-uniform l): v_readlane, not the LDS permute __shfl would use
__device__ __forceinline__ float owner_lane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double owner_lane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

template <typename T>
struct OwnerHp {
    T lr, regU, regI, regB, regC, gm, keepU, keepI, keepB, keepC; // keepX = 1 - lrate regX (the fp32 form of the update)
};

// One SGD update on the rows in registers: prediction, loss pieces, new values (h: hub row, x: spoke row; hc / sc their context-bias
// rows with lane c = condition c; hb / sb the scalar biases).  Shared by the wave-per-owner step and the team kernel's compute wave.
template <typename T, int MODEL, int VPL, int NCW, bool HUB_ITEM, bool STRICT>
__device__ __forceinline__ void owner_update(const OwnerRecT<NCW> &r, const OwnerHp<T> &hp, int k, T (&h)[VPL], T (&hc)[NCW], T &hb, T (&x)[VPL],
                                             T (&sc)[NCW], T &sb, T &sq_p, T &sq_q, T &sq_c, T &sq_e, T &sq_b) {
    using M = Traits<MODEL>;
    using S = Sides<MODEL, HUB_ITEM>;
    constexpr bool F32 = sizeof(T) == 4;
    // ---- prediction: hp.gm + bu + bj + (p.q + the context deviations); lane l adds the deviations of conditions l + 64 w
    bool sel[NCW];
    T term[NCW];
#pragma unroll
    for (int w = 0; w < NCW; ++w) {
        sel[w] = M::has_ctx && __builtin_amdgcn_inverse_ballot_w64(r.mask[w]);
        term[w] = (T)0;
        if (S::HC && S::SC) term[w] = HUB_ITEM ? hc[w] + sc[w] : sc[w] + hc[w]; // bic + buc
        else if (S::HC) term[w] = hc[w];
        else if (S::SC) term[w] = sc[w];
    }
    const T bu = HUB_ITEM ? sb : hb, bj = HUB_ITEM ? hb : sb;
    T pred = hp.gm;
    if (M::has_bu) pred += bu;
    if (M::has_bj) pred += bj;
    if constexpr (STRICT) {
        // the reference's operation order (fp64 state): DenseMatrix.rowMult sums m[f] * n[f] with f ascending (lane l holds
        // elements l VPL ...), then predict() adds the deviations condition by condition in ascending column order
        T prod[VPL];
#pragma unroll
        for (int v = 0; v < VPL; ++v) prod[v] = HUB_ITEM ? x[v] * h[v] : h[v] * x[v];
        T dot = (T)0;
        const int lanes = (k + VPL - 1) / VPL;
        for (int l = 0; l < lanes; ++l) {
#pragma unroll
            for (int v = 0; v < VPL; ++v)
                if (l * VPL + v < k) dot += owner_lane(prod[v], l);
        }
        pred += dot;
        if (M::has_ctx) {
#pragma unroll
            for (int w = 0; w < NCW; ++w)
                for (uint64_t m = r.mask[w]; m; m &= m - 1) pred += owner_lane(term[w], __builtin_ctzll(m));
        }
    } else {
        T part = (T)0;
#pragma unroll
        for (int v = 0; v < VPL; ++v) part = owner_fma(x[v], h[v], part);
        if (M::has_ctx) {
#pragma unroll
            for (int w = 0; w < NCW; ++w) part += sel[w] ? term[w] : (T)0;
        }
        pred += wave_total(part);
    }
    const T e = OwnerRating<T>::get(r) - pred;

    // ---- loss pieces (old values)
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const T pv = HUB_ITEM ? x[v] : h[v], qv = HUB_ITEM ? h[v] : x[v];
        sq_p = owner_fma(pv, pv, sq_p);
        sq_q = owner_fma(qv, qv, sq_q);
    }
    if (M::has_ctx) {
#pragma unroll
        for (int w = 0; w < NCW; ++w) {
            T cc = sq_c;
            if (S::HC) cc = owner_fma(hc[w], hc[w], cc);
            if (S::SC) cc = owner_fma(sc[w], sc[w], cc);
            sq_c = sel[w] ? cc : sq_c;
        }
    }
    sq_e = owner_fma(e, e, sq_e);
    if (M::has_bu) sq_b = owner_fma(bu, bu, sq_b);
    if (M::has_bj) sq_b = owner_fma(bj, bj, sq_b);
    // computed HERE: left alone, the compiler sinks these to the end of the round and keeps every step's old rows alive until then
    owner_settle(sq_p);
    owner_settle(sq_q);
    if (M::has_ctx) owner_settle(sq_c);
    owner_settle(sq_e);
    if (M::has_bu || M::has_bj) owner_settle(sq_b);

    // ---- updates (all from the old values)
    if constexpr (F32) {
        const T le = hp.lr * e;
        if (S::HB) hb = owner_fma(hp.keepB, hb, le);
        if (S::SB) sb = owner_fma(hp.keepB, sb, le);
#pragma unroll
        for (int w = 0; w < NCW; ++w) {
            if (S::HC) hc[w] = sel[w] ? owner_fma(hp.keepC, hc[w], le) : hc[w];
            if (S::SC) sc[w] = sel[w] ? owner_fma(hp.keepC, sc[w], le) : sc[w];
        }
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const T pv = HUB_ITEM ? x[v] : h[v], qv = HUB_ITEM ? h[v] : x[v];
            const T pn = owner_fma(le, qv, hp.keepU * pv);
            const T qn = owner_fma(le, pv, hp.keepI * qv);
            x[v] = HUB_ITEM ? pn : qn;
            h[v] = HUB_ITEM ? qn : pn;
        }
    } else {
        if (S::HB) hb = hb + hp.lr * (e - hp.regB * hb);
        if (S::SB) sb = sb + hp.lr * (e - hp.regB * sb);
#pragma unroll
        for (int w = 0; w < NCW; ++w) {
            if (S::HC) hc[w] = sel[w] ? hc[w] + hp.lr * (e - hp.regC * hc[w]) : hc[w];
            if (S::SC) sc[w] = sel[w] ? sc[w] + hp.lr * (e - hp.regC * sc[w]) : sc[w];
        }
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const T pv = HUB_ITEM ? x[v] : h[v], qv = HUB_ITEM ? h[v] : x[v];
            const T pn = pv + hp.lr * (e * qv - hp.regU * pv);
            const T qn = qv + hp.lr * (e * pv - hp.regI * qv);
            x[v] = HUB_ITEM ? pn : qn;
            h[v] = HUB_ITEM ? qn : pn;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// team form: the hottest owners as THREE wavefronts each (one workgroup = one owner)
// ---------------------------------------------------------------------------------------------
// A lone wavefront issues a dependent instruction every ~10 cycles and an independent one every ~5.4 (tools/micro/issue_rate.hip), so the
// pace of the owner of a hot row is the instruction count of its step.  For the owners whose list is long enough to bound the epoch,
// the step is split over three wavefronts of one workgroup, on three SIMDs, that pass the spoke rows through a ring in LDS:
//   loader  (wave 1): reads the records ahead (D deep, as the one-wave form does), validates the tags -- polling when a predecessor is
//                     late -- and puts the row, context biases and bias of entry c into ring slot c mod R;  n_ready = c + 1
//   compute (wave 0): hub row in registers; takes slot c, runs owner_update, puts the new values back into the slot;  n_done = c + 1
//   storer  (wave 2): takes the new values, tags them want + 1 and writes the record through;  n_stored = c + 1 (the slot is free again)
// The three counters live in LDS, each written by one wave: LDS operations of a wave execute in order, so "data, then counter" needs no
// fence, and the readers poll.  What is left on the compute wave is the arithmetic chain itself.  Any list is handled correctly (a hub
// switch loads the hub row synchronously), but the form only pays for single-hub lists: cmi_set_ratings gives it to the longest ones.
static const int OWNER_TEAM_RING = 32;

template <typename T, int VPL>
struct TeamVec {
    typedef T type __attribute__((ext_vector_type(VPL)));
};
template <typename T>
struct TeamVec<T, 1> {
    typedef T type;
};
// LDS pointers keep their address space explicitly: through a generic pointer these accesses would become flat_load / flat_store, which
// travel the vector-memory path and count on both vmcnt and lgkmcnt
typedef __attribute__((address_space(3))) unsigned char lds_u8;
template <typename T, int VPL>
__device__ __forceinline__ void team_get(const lds_u8 *p, T (&v)[VPL]) {
    typedef typename TeamVec<T, VPL>::type V;
    const V t = *(const volatile __attribute__((address_space(3))) V *)p;
    if constexpr (VPL == 1) v[0] = t;
    else {
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[i] = t[i];
    }
}
template <typename T, int VPL>
__device__ __forceinline__ void team_put(lds_u8 *p, const T (&v)[VPL]) {
    typedef typename TeamVec<T, VPL>::type V;
    V t;
    if constexpr (VPL == 1) t = v[0];
    else {
#pragma unroll
        for (int i = 0; i < VPL; ++i) t[i] = v[i];
    }
    *(volatile __attribute__((address_space(3))) V *)p = t;
}

// A zero the compiler cannot see through, derived from a value read from LDS: an index that adds it cannot be formed -- and the scalar
// load that uses it cannot be issued -- before that LDS read has returned.  Scalar loads and LDS operations share one counter, which can
// only be waited down to zero while a scalar load is outstanding; issued after the step's LDS reads, the load of the next list entry has
// the whole arithmetic of the step to complete in instead of being waited for together with them.
__device__ __forceinline__ int team_zero_after(int bits) {
    int z;
    asm volatile("s_and_b32 %0, %1, 0" : "=s"(z) : "s"(__builtin_amdgcn_readfirstlane(bits)) : "scc"); // opaque: `bits & 0` would be folded away
    return z;
}
__device__ __forceinline__ int team_after(float v) { return team_zero_after(__float_as_int(v)); }
__device__ __forceinline__ int team_after(double v) { return team_zero_after(__double2loint(v)); }

template <typename T, int MODEL, int VPL, int NCW, bool HUB_ITEM>
__host__ __device__ constexpr int team_slot_bytes() { // row | context biases | bias (16)
    return 64 * VPL * (int)sizeof(T) + ((HUB_ITEM ? Traits<MODEL>::has_uc : Traits<MODEL>::has_ic) ? 64 * NCW * (int)sizeof(T) : 0) + 16;
}

template <typename T, int MODEL, int VPL, int NCW, int D, bool HUB_ITEM>
__device__ __forceinline__ void owner_team(const SgdArgs<T> &a, const OwnerRecT<NCW> *__restrict__ recs, int len, int w, __amdgpu_buffer_rsrc_t rs,
                                           const OwnerHp<T> &hp, int *error, int64_t pos0) {
    typedef OwnerRecT<NCW> OwnerRec;
    using S = Sides<MODEL, HUB_ITEM>;
    constexpr int NW = Tagged<T>::NW, R = OWNER_TEAM_RING;
    constexpr int XB = 64 * VPL * (int)sizeof(T), CB = S::SC ? 64 * NCW * (int)sizeof(T) : 0, SLOT = team_slot_bytes<T, MODEL, VPL, NCW, HUB_ITEM>();
    constexpr int OFF_SB = XB + CB;
    static_assert(D <= R, "the loader never waits for ring space on the inert entries that complete its last round");
    extern __shared__ __attribute__((aligned(16))) unsigned char team_lds[];
    volatile __attribute__((address_space(3))) uint32_t *ctr =
        (volatile __attribute__((address_space(3))) uint32_t *)team_lds; // [0] n_ready, [1] n_done, [2] n_stored
    lds_u8 *ring = (lds_u8 *)team_lds + 64;
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const int k = a.k;
    if (threadIdx.x < 3) ctr[threadIdx.x] = 0u;
    __syncthreads();
    if (role == 3) return;

    if (role == 1) { // ---------------- loader
        OwnerSlot<T, VPL, NCW> slot[D];
#pragma unroll
        for (int d = 0; d < D; ++d) owner_load_spoke<T, MODEL, VPL, NCW, HUB_ITEM>(rs, (int)recs[d].off, lane, slot[d]);
        OwnerRec r_run = recs[0], r_ahead = recs[D];
        uint32_t freed = 0; // entries the storer is known to have taken out of the ring
        auto step = [&](int c, OwnerSlot<T, VPL, NCW> &s) {
            const OwnerRec r = r_run, p = r_ahead;
            r_run = recs[c + 1];
            r_ahead = recs[c + 1 + D];
            if (__builtin_expect(!(r.flags & OWN_NOP), 1)) {
                if (__builtin_expect((uint32_t)c - freed >= (uint32_t)R, 0)) {
                    unsigned spins = 0;
                    while ((uint32_t)c - (freed = ctr[2]) >= (uint32_t)R) {
                        if (w == 0 && lane == 0 && spins == 0) atomicAdd(error + 2, 1); // statistics: the ring was full
                        __builtin_amdgcn_s_sleep(1);
                        if (owner_spin_expired(spins, error, lane)) break;
                    }
                }
                lds_u8 *sl = ring + (c % R) * SLOT;
                T sbv[1];
                sbv[0] = (T)0;
                if (!(r.flags & OWN_SPK_FWD)) {
                    if (__builtin_expect(!__all(owner_spoke_ok<T, MODEL, VPL, NCW, HUB_ITEM>(s, r.want + a.owner_tag0)), 0)) {
                        unsigned spins = 0;
                        while (true) {
                            __builtin_amdgcn_s_sleep(4);
                            owner_load_spoke<T, MODEL, VPL, NCW, HUB_ITEM>(rs, (int)r.off, lane, s);
                            if (__all(owner_spoke_ok<T, MODEL, VPL, NCW, HUB_ITEM>(s, r.want + a.owner_tag0))) break;
                            if (owner_spin_expired(spins, error, lane)) break;
                        }
                    }
                    T xv[VPL], one[1];
#pragma unroll
                    for (int v = 0; v < VPL; ++v) xv[v] = owner_elem(s.xw, v, (T)0);
                    team_put<T, VPL>(sl + lane * (VPL * (int)sizeof(T)), xv);
                    CMI_TR(if constexpr (VPL == 1 && NCW == 1) {
                        tr_put(a.trace, pos0 + c, 8, lane, xv[0]);
                        if (S::SB) tr_put(a.trace, pos0 + c, 9, lane, owner_elem(s.bw, 0, (T)0));
                    })
                    if (S::SC) {
#pragma unroll
                        for (int cw = 0; cw < NCW; ++cw) {
                            one[0] = owner_elem(s.cw[cw], 0, (T)0);
                            team_put<T, 1>(sl + XB + (64 * cw + lane) * (int)sizeof(T), one);
                        }
                    }
                    if (S::SB) sbv[0] = owner_elem(s.bw, 0, (T)0);
                }
                if (lane == 0) { // the counter after the slot's data: LDS operations of a wave execute in order
                    if (S::SB && !(r.flags & OWN_SPK_FWD)) team_put<T, 1>(sl + OFF_SB, sbv);
                    ctr[0] = (uint32_t)c + 1u;
                }
            }
            owner_load_spoke<T, MODEL, VPL, NCW, HUB_ITEM>(rs, (int)p.off, lane, s);
        };
        for (int base = 0; base < len; base += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) step(base + d, slot[d]);
        }
        return;
    }

    if (role == 2) { // ---------------- storer
        uint32_t done = 0;
        OwnerRec r_next = recs[0];
        for (int c = 0; c < len; ++c) {
            const OwnerRec r = r_next;
            if (__builtin_expect((uint32_t)c >= done, 0)) {
                unsigned spins = 0;
                while ((uint32_t)c >= (done = ctr[1])) {
                    __builtin_amdgcn_s_sleep(1);
                    if (owner_spin_expired(spins, error, lane)) break;
                }
            }
            const lds_u8 *sl = ring + (c % R) * SLOT;
            T xv[VPL], cv[NCW][1], bv[1];
            team_get<T, VPL>(sl + lane * (VPL * (int)sizeof(T)), xv);
            if (S::SC) {
#pragma unroll
                for (int cw = 0; cw < NCW; ++cw) team_get<T, 1>(sl + XB + (64 * cw + lane) * (int)sizeof(T), cv[cw]);
            }
            if (S::SB) team_get<T, 1>(sl + OFF_SB, bv);
            {
                T last = xv[0];
                if (S::SC) last = cv[NCW - 1][0];
                if (S::SB) last = bv[0];
                r_next = recs[c + 1 + team_after(last)];
            }
            const uint32_t tag = r.want + a.owner_tag0 + 1u;
            uint32_t ow[VPL * NW * 2], oc[NW * 2], ob[NW * 2];
            CMI_TR(if constexpr (VPL == 1 && NCW == 1) {
                tr_put(a.trace, pos0 + c, 10, lane, xv[0]);
                if (S::SB) tr_put(a.trace, pos0 + c, 11, lane, bv[0]);
            })
#pragma unroll
            for (int v = 0; v < VPL; ++v) owner_pack(ow, v, xv[v], tag);
            owner_st_words(rs, lane * (VPL * NW * 8), (int)r.off, ow);
            if (S::SC) {
#pragma unroll
                for (int cw = 0; cw < NCW; ++cw) {
                    owner_pack(oc, 0, cv[cw][0], tag);
                    owner_st_words(rs, (64 * VPL + 64 * cw) * NW * 8 + lane * (NW * 8), (int)r.off, oc);
                }
            }
            if (S::SB) {
                owner_pack(ob, 0, bv[0], tag);
                owner_st_words(rs, (64 * VPL + (S::SC ? 64 * NCW : 0)) * NW * 8, (int)r.off, ob);
            }
            if (lane == 0) ctr[2] = (uint32_t)c + 1u; // after the reads of the slot (in order): the loader may refill it
        }
        return;
    }

    // ---------------- compute
    T h[VPL], hc[NCW], hb = (T)0, x[VPL], sc[NCW], sb = (T)0;
#pragma unroll
    for (int v = 0; v < VPL; ++v) h[v] = x[v] = (T)0;
#pragma unroll
    for (int cw = 0; cw < NCW; ++cw) hc[cw] = sc[cw] = (T)0;
    T sq_p = (T)0, sq_q = (T)0, sq_c = (T)0, sq_e = (T)0, sq_b = (T)0;
    double acc = 0.0;
    uint32_t ready = 0;
    OwnerRec r_next = recs[0];
    for (int c = 0; c < len; ++c) {
        const OwnerRec r = r_next;
        // EVERY entry waits for the loader (also those whose row is taken over in registers): the loader's ring arithmetic counts on
        // the compute wave never being ahead of it.  Only the owner's head entry is ever waited for.
        if (__builtin_expect((uint32_t)c >= ready, 0)) {
            unsigned spins = 0;
            while ((uint32_t)c >= (ready = ctr[0])) {
                if (w == 0 && lane == 0 && spins == 0) atomicAdd(error + 1, 1); // statistics: the compute wave waited for the loader
                __builtin_amdgcn_s_sleep(1);
                if (owner_spin_expired(spins, error, lane)) break;
            }
        }
        lds_u8 *sl = ring + (c % R) * SLOT;
        if (__builtin_expect(!(r.flags & OWN_HUB_FWD), 0)) { // a single-hub list: once
            T hq[VPL], hcq[NCW], hbq = (T)0;
#pragma unroll
            for (int cw = 0; cw < NCW; ++cw) hcq[cw] = (T)0;
#ifdef CMI_VAR_HUB_COHERENT
            { // experiment: the team's hub row through system-scope (sc0 sc1) loads
                const T *row = (HUB_ITEM ? a.Q : a.P) + (size_t)r.hub * k;
#pragma unroll
                for (int v = 0; v < VPL; ++v) hq[v] = __hip_atomic_load(row + min(lane * VPL + v, k - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (S::HC) {
#pragma unroll
                    for (int cw = 0; cw < NCW; ++cw)
                        hcq[cw] = __hip_atomic_load((HUB_ITEM ? a.icBias : a.ucBias) + (size_t)r.hub * a.n_conds + min(lane + 64 * cw, a.n_conds - 1),
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                if (S::HB) hbq = __hip_atomic_load((HUB_ITEM ? a.itemBias : a.userBias) + r.hub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
#else
            owner_load_hub<T, MODEL, VPL, NCW, HUB_ITEM>(a, r.hub, lane, k, hq, hcq, hbq);
#endif
#pragma unroll
            for (int v = 0; v < VPL; ++v) h[v] = lane * VPL + v < k ? hq[v] : (T)0;
            if (S::HC) {
#pragma unroll
                for (int cw = 0; cw < NCW; ++cw) hc[cw] = lane + 64 * cw < a.n_conds ? hcq[cw] : (T)0;
            }
            if (S::HB) hb = hbq;
        }
        if (!(r.flags & OWN_SPK_FWD)) {
            T one[1];
            team_get<T, VPL>(sl + lane * (VPL * (int)sizeof(T)), x);
            if (S::SC) {
#pragma unroll
                for (int cw = 0; cw < NCW; ++cw) {
                    team_get<T, 1>(sl + XB + (64 * cw + lane) * (int)sizeof(T), one);
                    sc[cw] = one[0];
                }
            }
            if (S::SB) {
                team_get<T, 1>(sl + OFF_SB, one);
                sb = one[0];
            }
        }
        {
            T last = x[0]; // the value of the step's LAST LDS read: every read has returned before the next entry is requested
            if (S::SC) last = sc[NCW - 1];
            if (S::SB) last = sb;
            r_next = recs[c + 1 + team_after(last)];
        }
        CMI_TR(if constexpr (VPL == 1 && NCW == 1) {
            tr_put(a.trace, pos0 + c, 0, lane, h[0]);
            tr_put(a.trace, pos0 + c, 1, lane, x[0]);
            tr_put(a.trace, pos0 + c, 2, lane, hc[0]);
            const double meta = lane == 0 ? (double)sb : lane == 1 ? (double)r.off : lane == 2 ? (double)r.hub : lane == 3 ? (double)r.want
                              : lane == 4 ? (double)r.flags : lane == 5 ? 1.0 : lane == 6 ? (double)a.owner_tag0 : lane == 7 ? (double)hb : 0.0;
            tr_put(a.trace, pos0 + c, 3, lane, meta);
        })
        owner_update<T, MODEL, VPL, NCW, HUB_ITEM, false>(r, hp, k, h, hc, hb, x, sc, sb, sq_p, sq_q, sq_c, sq_e, sq_b);
        CMI_TR(if constexpr (VPL == 1 && NCW == 1) {
            tr_put(a.trace, pos0 + c, 4, lane, h[0]);
            tr_put(a.trace, pos0 + c, 5, lane, x[0]);
            tr_put(a.trace, pos0 + c, 6, lane, hc[0]);
            tr_put(a.trace, pos0 + c, 7, lane, lane == 0 ? sb : lane == 7 ? hb : (T)0);
        })
        {
            T one[1];
            team_put<T, VPL>(sl + lane * (VPL * (int)sizeof(T)), x);
            if (S::SC) {
#pragma unroll
                for (int cw = 0; cw < NCW; ++cw) {
                    one[0] = sc[cw];
                    team_put<T, 1>(sl + XB + (64 * cw + lane) * (int)sizeof(T), one);
                }
            }
            if (lane == 0) {
                if (S::SB) {
                    one[0] = sb;
                    team_put<T, 1>(sl + OFF_SB, one);
                }
                ctr[1] = (uint32_t)c + 1u;
            }
        }
        if (__builtin_expect(r.flags & OWN_HUB_STORE, 0)) {
            T *row = (HUB_ITEM ? a.Q : a.P) + (size_t)r.hub * k;
#ifdef CMI_VAR_HUB_COHERENT
#define CMI_HUB_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#else
#define CMI_HUB_ST(p, v) (*(p) = (v))
#endif
#pragma unroll
            for (int v = 0; v < VPL; ++v)
                if (lane * VPL + v < k) CMI_HUB_ST(row + lane * VPL + v, h[v]);
            if (S::HC) {
#pragma unroll
                for (int cw = 0; cw < NCW; ++cw)
                    if (lane + 64 * cw < a.n_conds) CMI_HUB_ST((HUB_ITEM ? a.icBias : a.ucBias) + (size_t)r.hub * a.n_conds + lane + 64 * cw, hc[cw]);
            }
            if (S::HB && lane == 0) CMI_HUB_ST((HUB_ITEM ? a.itemBias : a.userBias) + r.hub, hb);
#undef CMI_HUB_ST
        }
        if ((c & 15) == 15) { // flush the float partial sums of squares into the double accumulator
            acc += (double)hp.regU * (double)sq_p + (double)hp.regI * (double)sq_q + (double)hp.regC * (double)sq_c;
            if (lane == 0) acc += (double)sq_e + (double)hp.regB * (double)sq_b;
            sq_p = sq_q = sq_c = sq_e = sq_b = (T)0;
        }
    }
    acc += (double)hp.regU * (double)sq_p + (double)hp.regI * (double)sq_q + (double)hp.regC * (double)sq_c;
    if (lane == 0) acc += (double)sq_e + (double)hp.regB * (double)sq_b;
    acc = wave_sum64(acc);
    if (lane == 0) a.loss_part[w] = acc;
}

// A lone wavefront issues a dependent instruction every ~10 cycles and an independent one every ~5.4 (tools/micro/issue_rate.hip), so
// the hottest owner's pace is the INSTRUCTION COUNT of a step: the step is written for few instructions -- condition masks go straight into v_cndmask as lane masks (inverse ballot),
// the wave sum leaves through one readlane, the fp32 update is two fused operations per element (new = (1 - lrate reg) old + (lrate
// e) other: the same value as old + lrate (e other - reg old) up to rounding; the fp64 kernel keeps the reference's expression and
// operation order), the loss is accumulated per lane and reduced once per owner.
template <typename T, int MODEL, int VPL, int NCW, int D, bool HUB_ITEM, bool STRICT>
__global__ __launch_bounds__(256, 1) void sgd_owner(SgdArgs<T> a, const OwnerRecT<NCW> *__restrict__ recs, const int64_t *__restrict__ own_off,
                                                    gran_t *tagged, int *error, int n_owners, int n_team) {
    typedef OwnerRecT<NCW> OwnerRec;
    constexpr int NW = Tagged<T>::NW;
    using S = Sides<MODEL, HUB_ITEM>;
    static_assert(MODEL != CAMF_C, "CAMF_C has no owner schedule (shared condBias)");
    constexpr bool F32 = sizeof(T) == 4;
    static_assert(!STRICT || !F32, "the strict form is the fp64 reference arithmetic");
    const int lane = threadIdx.x & 63;
    // workgroups [0, n_team): one owner each, as a team of three wavefronts; the rest: four owners each, one per wavefront
    const bool team = !STRICT && (int)blockIdx.x < n_team;
    const int w = team ? (int)blockIdx.x : __builtin_amdgcn_readfirstlane(n_team + ((int)blockIdx.x - n_team) * 4 + (int)(threadIdx.x >> 6));
    if (w >= n_owners) return;
    const int k = a.k;
    const HParams hpd = *a.hp;
    const T lr = (T)hpd.lr, regU = (T)hpd.regU, regI = (T)hpd.regI, regB = (T)hpd.regB, regC = (T)hpd.regC, gm = (T)hpd.gm;
    const OwnerHp<T> hp{lr, regU, regI, regB, regC, gm, (T)1 - lr * regU, (T)1 - lr * regI, (T)1 - lr * regB, (T)1 - lr * regC};
    // the owner's list, followed by 2 OWNER_DEPTH_MAX inert entries (OWN_NOP: the step computes nothing and stores to / reads ahead from
    // the owner's dummy record behind the table): enough to complete the last round and to read D + 1 entries past it
    static_assert(D <= OWNER_DEPTH_MAX, "list padding");
    const int64_t c0 = own_off[w] + (int64_t)w * (2 * OWNER_DEPTH_MAX);
    const int len = (int)(own_off[w + 1] - own_off[w]);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tagged, 0, 0xffffffff, 0x00020000);
    if (len == 0) {
        if (lane == 0 && (!team || threadIdx.x < 64)) a.loss_part[w] = 0.0;
        return;
    }
    recs += c0;
    if constexpr (!STRICT) {
        if (team) {
            owner_team<T, MODEL, VPL, NCW, D, HUB_ITEM>(a, recs, len, w, rs, hp, error, c0);
            return;
        }
    }

    OwnerSlot<T, VPL, NCW> slot[D];
    // current rows (registers): hub row / context-bias row / bias, and the spoke's after its latest update
    T h[VPL], hc[NCW], hb = (T)0, x[VPL], sc[NCW], sb = (T)0;
#pragma unroll
    for (int v = 0; v < VPL; ++v) h[v] = x[v] = (T)0;
#pragma unroll
    for (int cw = 0; cw < NCW; ++cw) hc[cw] = sc[cw] = (T)0;
    // loss: per lane sums of squares, scaled and reduced once at the end (flushed to double every round of D steps)
    T sq_p = (T)0, sq_q = (T)0, sq_c = (T)0, sq_e = (T)0, sq_b = (T)0;
    double acc = 0.0;
    int n_late = 0, n_polls = 0;

    // The spoke record is ALWAYS read ahead (a fixed number of memory operations per step keeps the compiler's counted waits deep);
    // the hub side only when the step will not take it over in registers.
    auto prefetch = [&](const OwnerRec &r, OwnerSlot<T, VPL, NCW> &s) {
        owner_load_spoke<T, MODEL, VPL, NCW, HUB_ITEM>(rs, (int)r.off, lane, s);
        if (__builtin_expect(!(r.flags & (OWN_HUB_FWD | OWN_HUB_LATE)), 0)) owner_load_hub<T, MODEL, VPL, NCW, HUB_ITEM>(a, r.hub, lane, k, s.hq, s.hc, s.hb);
    };

#pragma unroll
    for (int d = 0; d < D; ++d) prefetch(recs[d], slot[d]);
    // list entries arrive one step ahead of their use (scalar loads): the step's own, and the one it reads ahead for
    OwnerRec r_run = recs[0], r_ahead = recs[D];

    // one step of the list: tuple c, whose read-ahead sits in s; ends by reading ahead for tuple c + D into the same registers
    auto step = [&](int c, OwnerSlot<T, VPL, NCW> &s) {
        const OwnerRec r = r_run, p = r_ahead;
        r_run = recs[c + 1];
        r_ahead = recs[c + 1 + D];
        if (__builtin_expect(!(r.flags & OWN_NOP), 1)) { // OWN_NOP: past the end of the list; only the fixed store / read-ahead sequence runs, on a dummy record

        // ---- hub side: registers (same row as the previous step) | read ahead | re-read now (written < D steps ago)
        if (__builtin_expect(!(r.flags & OWN_HUB_FWD), 0)) { // (unlikely: the layout that matters is the hottest owner's)
            if (r.flags & OWN_HUB_LATE) {
                owner_load_hub<T, MODEL, VPL, NCW, HUB_ITEM>(a, r.hub, lane, k, s.hq, s.hc, s.hb);
#pragma unroll
                for (int v = 0; v < VPL; ++v) owner_settle(s.hq[v]);
                if (S::HC) {
#pragma unroll
                    for (int cw = 0; cw < NCW; ++cw) owner_settle(s.hc[cw]);
                }
                owner_settle(s.hb);
            }
#pragma unroll
            for (int v = 0; v < VPL; ++v) h[v] = lane * VPL + v < k ? s.hq[v] : (T)0;
            if (S::HC) {
#pragma unroll
                for (int cw = 0; cw < NCW; ++cw) hc[cw] = lane + 64 * cw < a.n_conds ? s.hc[cw] : (T)0;
            }
            if (S::HB) hb = s.hb;
        }
        // ---- spoke side: registers | the record read ahead, if every granule carries the tag | poll
        if (!(r.flags & OWN_SPK_FWD)) {
            if (__builtin_expect(!__all(owner_spoke_ok<T, MODEL, VPL, NCW, HUB_ITEM>(s, r.want + a.owner_tag0)), 0)) {
                unsigned spins = 0;
                while (true) { // the predecessor has not written the record yet (or was in the middle of it)
                    __builtin_amdgcn_s_sleep(4);
                    owner_load_spoke<T, MODEL, VPL, NCW, HUB_ITEM>(rs, (int)r.off, lane, s);
                    if (__all(owner_spoke_ok<T, MODEL, VPL, NCW, HUB_ITEM>(s, r.want + a.owner_tag0))) break;
                    if (owner_spin_expired(spins, error, lane)) break;
                }
                n_late += 1; // statistics: records that were not ready when their step came, and the polls spent on them
                n_polls += (int)spins + 1;
            }
#pragma unroll
            for (int v = 0; v < VPL; ++v) x[v] = owner_elem(s.xw, v, (T)0);
            if (S::SC) {
#pragma unroll
                for (int cw = 0; cw < NCW; ++cw) sc[cw] = owner_elem(s.cw[cw], 0, (T)0);
            }
            if (S::SB) sb = owner_elem(s.bw, 0, (T)0);
        }

        CMI_TR(if constexpr (VPL == 1 && NCW == 1) {
            tr_put(a.trace, c0 + c, 0, lane, h[0]);
            tr_put(a.trace, c0 + c, 1, lane, x[0]);
            tr_put(a.trace, c0 + c, 2, lane, hc[0]);
            const double meta = lane == 0 ? (double)sb : lane == 1 ? (double)r.off : lane == 2 ? (double)r.hub : lane == 3 ? (double)r.want
                              : lane == 4 ? (double)r.flags : lane == 5 ? 1.0 : lane == 6 ? (double)a.owner_tag0 : lane == 7 ? (double)hb : 0.0;
            tr_put(a.trace, c0 + c, 3, lane, meta);
        })
        owner_update<T, MODEL, VPL, NCW, HUB_ITEM, STRICT>(r, hp, k, h, hc, hb, x, sc, sb, sq_p, sq_q, sq_c, sq_e, sq_b);
        CMI_TR(if constexpr (VPL == 1 && NCW == 1) {
            tr_put(a.trace, c0 + c, 4, lane, h[0]);
            tr_put(a.trace, c0 + c, 5, lane, x[0]);
            tr_put(a.trace, c0 + c, 6, lane, hc[0]);
            tr_put(a.trace, c0 + c, 7, lane, lane == 0 ? sb : lane == 7 ? hb : (T)0);
        })
        } // !OWN_NOP

        // ---- the spoke record goes back with the next tag (always: one store sequence per step); the hub side when the next
        //      step of the list does not take it over in registers
        {
            const uint32_t tag = r.want + a.owner_tag0 + 1u;
            uint32_t ow[VPL * NW * 2], oc[NW * 2], ob[NW * 2];
#pragma unroll
            for (int v = 0; v < VPL; ++v) owner_pack(ow, v, x[v], tag);
            owner_st_words(rs, lane * (VPL * NW * 8), (int)r.off, ow);
            if (S::SC) {
#pragma unroll
                for (int cw = 0; cw < NCW; ++cw) {
                    owner_pack(oc, 0, sc[cw], tag);
                    owner_st_words(rs, (64 * VPL + 64 * cw) * NW * 8 + lane * (NW * 8), (int)r.off, oc);
                }
            }
            if (S::SB) {
                owner_pack(ob, 0, sb, tag);
                owner_st_words(rs, (64 * VPL + (S::SC ? 64 * NCW : 0)) * NW * 8, (int)r.off, ob);
            }
        }
        if (__builtin_expect(r.flags & OWN_HUB_STORE, 0)) {
            T *row = (HUB_ITEM ? a.Q : a.P) + (size_t)r.hub * k;
#pragma unroll
            for (int v = 0; v < VPL; ++v)
                if (lane * VPL + v < k) row[lane * VPL + v] = h[v];
            if (S::HC) {
#pragma unroll
                for (int cw = 0; cw < NCW; ++cw)
                    if (lane + 64 * cw < a.n_conds) (HUB_ITEM ? a.icBias : a.ucBias)[(size_t)r.hub * a.n_conds + lane + 64 * cw] = hc[cw];
            }
            if (S::HB && lane == 0) (HUB_ITEM ? a.itemBias : a.userBias)[r.hub] = hb;
        }

        // ---- read ahead for the step D places down the list (after this step's stores: program order covers a row of this
        //      owner that comes round again); past the end of the list the last entry is read again, to no effect
        prefetch(p, s);
    };
    auto flush_loss = [&]() {
        acc += (double)regU * (double)sq_p + (double)regI * (double)sq_q + (double)regC * (double)sq_c;
        if (lane == 0) acc += (double)sq_e + (double)regB * (double)sq_b; // the uniform terms, once
        sq_p = sq_q = sq_c = sq_e = sq_b = (T)0;
    };

    // Rounds of D steps with no exit inside the round (an exit edge from the middle of a round to the loop latch would make the
    // shortest path to the next round's first wait a few operations long, and the compiler would size every wait for it): the list
    // is followed by inert entries (OWN_NOP) up to a whole number of rounds.
    for (int base = 0; base < len; base += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) step(base + d, slot[d]);
        flush_loss();
    }
    acc = wave_sum64(acc);
    if (lane == 0) {
        a.loss_part[w] = acc;
        if (n_late) { // statistics (CMI_OWNER_STATS): the busiest owner is owner 0
            if (w == 0) {
                atomicAdd(error + 1, n_late);
                atomicAdd(error + 2, n_polls);
            }
            atomicAdd(error + 3, n_late);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// Read-ahead distance.  Loads and write-through stores retire through ONE in-order counter (vmcnt, at most 63 outstanding), so the wait
// for a record read D steps ago also waits for the stores issued before it, and a write-through store takes microseconds to be
// acknowledged: the step time cannot go below (store latency) / D.  D = 16 where a step issues 4 memory operations (fp32, k <= 128, at
// most 64 conditions: 15 x 4 = 60 in flight), 8 where it issues up to 8, 4 for the widest context rows (6 x 64 conditions).
template <typename T, int VPL, int NCW>
struct OwnerDepth {
    static constexpr int D = (sizeof(T) == 4 && VPL <= 2 && NCW == 1) ? 16 : (NCW <= 2 ? 8 : 4);
};

// 64-condition words of a context-bias row: 1, 2 or 6 (<= 64, <= 128, <= 384 conditions); models without context: 1
int owner_mask_words(int model, int n_conds) {
    const bool has_ctx = model != BIASEDMF && model != PMF;
    if (!has_ctx || n_conds <= 64) return 1;
    return n_conds <= 128 ? 2 : 6;
}

bool has_owner_path(int model, int k, int n_conds, bool f64, bool strict) {
    if (strict && !f64) return false; // the strict form is the fp64 reference arithmetic
    if (model != BIASEDMF && model != PMF && model != CAMF_CI && model != CAMF_CU && model != CAMF_CUCI) return false;
    if (k < 1 || k > (f64 ? 128 : 256)) return false;
    const bool has_ctx = model != BIASEDMF && model != PMF;
    if (has_ctx && n_conds > 384) return false; // lane l carries conditions l, l + 64, ..., l + 320
    return true;
}
int owner_depth() { return OWNER_DEPTH_MAX; }

int64_t owner_record_stride(int model, int k, int n_conds, bool f64, bool hub_is_item) {
    const bool has_ic = model == CAMF_CI || model == CAMF_CUCI, has_uc = model == CAMF_CU || model == CAMF_CUCI;
    const bool has_bu = model == BIASEDMF || model == CAMF_CI, has_bj = model == BIASEDMF || model == CAMF_CU;
    const bool sc = hub_is_item ? has_uc : has_ic, sb = hub_is_item ? has_bu : has_bj;
    return owner_record_granules(owner_vpl(k), sc ? owner_mask_words(model, n_conds) : 0, sb, f64 ? 2 : 1);
}

// what the launch needs to know about one instantiation
struct OwnerKernel {
    void *fn;
    size_t team_lds;
};
template <typename T, int MODEL, int VPL, int NCW>
static OwnerKernel owner_kernel_hub(bool hub_is_item, bool strict) {
    constexpr int D = OwnerDepth<T, VPL, NCW>::D;
    const size_t lds = 64 + (size_t)OWNER_TEAM_RING *
                                (size_t)(hub_is_item ? team_slot_bytes<T, MODEL, VPL, NCW, true>() : team_slot_bytes<T, MODEL, VPL, NCW, false>());
    if constexpr (sizeof(T) == 8) {
        if (strict)
            return {hub_is_item ? (void *)sgd_owner<T, MODEL, VPL, NCW, D, true, true> : (void *)sgd_owner<T, MODEL, VPL, NCW, D, false, true>, 0};
    }
    return {hub_is_item ? (void *)sgd_owner<T, MODEL, VPL, NCW, D, true, false> : (void *)sgd_owner<T, MODEL, VPL, NCW, D, false, false>, lds};
}
template <typename T, int MODEL, int NCW>
static OwnerKernel owner_kernel_k(int k, bool hub_is_item, bool strict) {
    if (k <= 64) return owner_kernel_hub<T, MODEL, 1, NCW>(hub_is_item, strict);
    if (k <= 128) return owner_kernel_hub<T, MODEL, 2, NCW>(hub_is_item, strict);
    if constexpr (sizeof(T) == 4) return owner_kernel_hub<float, MODEL, 4, NCW>(hub_is_item, strict);
    return {nullptr, 0};
}
template <typename T, int MODEL>
static OwnerKernel owner_kernel_ctx(int ncw, int k, bool hub_is_item, bool strict) {
    if (ncw == 1) return owner_kernel_k<T, MODEL, 1>(k, hub_is_item, strict);
    if (ncw == 2) return owner_kernel_k<T, MODEL, 2>(k, hub_is_item, strict);
    return owner_kernel_k<T, MODEL, 6>(k, hub_is_item, strict);
}
template <typename T>
static OwnerKernel owner_kernel(int model, int n_conds, int k, bool hub_is_item, bool strict) {
    const int ncw = owner_mask_words(model, n_conds);
    switch (model) {
    case BIASEDMF: return owner_kernel_k<T, BIASEDMF, 1>(k, hub_is_item, strict);
    case PMF: return owner_kernel_k<T, PMF, 1>(k, hub_is_item, strict);
    case CAMF_CI: return owner_kernel_ctx<T, CAMF_CI>(ncw, k, hub_is_item, strict);
    case CAMF_CU: return owner_kernel_ctx<T, CAMF_CU>(ncw, k, hub_is_item, strict);
    case CAMF_CUCI: return owner_kernel_ctx<T, CAMF_CUCI>(ncw, k, hub_is_item, strict);
    }
    return {nullptr, 0};
}

// Owners = wavefronts that are resident together (a waiting owner must never keep a runnable one off the chip).
int owner_grid_waves(int device, int model, int n_conds, int k, bool f64, bool hub_is_item) {
    void *fn = f64 ? owner_kernel<double>(model, n_conds, k, hub_is_item, false).fn : owner_kernel<float>(model, n_conds, k, hub_is_item, false).fn;
    if (!fn) return 0;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) return 0;
    if (per_cu > 1) per_cu = 1; // __launch_bounds__(256, 1): one owner per SIMD, the whole register file for its read-ahead
    if (const char *env = cmi_exp_env("CMI_OWNER_BLOCKS_PER_CU")) {
        const int v = atoi(env);
        if (v >= 1 && v < per_cu) per_cu = v;
    }
    return cus * per_cu * 4;
}

template <typename T>
hipError_t launch_owner_epoch(const SgdArgs<T> &a, int model, bool hub_is_item, bool strict, const void *recs, const int64_t *own_off,
                              int n_owners, int n_team, void *tagged, int64_t stride, int n_spokes, int *error, uint32_t tag0, hipStream_t s) {
    const OwnerKernel kn = owner_kernel<T>(model, a.n_conds, a.k, hub_is_item, strict);
    if (!kn.fn) return hipErrorInvalidValue;
    const bool has_ic = model == CAMF_CI || model == CAMF_CUCI, has_uc = model == CAMF_CU || model == CAMF_CUCI;
    const bool has_bu = model == BIASEDMF || model == CAMF_CI, has_bj = model == BIASEDMF || model == CAMF_CU;
    const bool sc = hub_is_item ? has_uc : has_ic, sb = hub_is_item ? has_bu : has_bj;
    T *rows = hub_is_item ? a.P : a.Q;
    T *ctx = sc ? (hub_is_item ? a.ucBias : a.icBias) : nullptr;
    T *bias = sb ? (hub_is_item ? a.userBias : a.itemBias) : nullptr;
    const int ncs = sc ? a.n_conds : 0, vpl = owner_vpl(a.k), ncw = owner_mask_words(model, a.n_conds);
    const int pass_blocks = (int)std::min<int64_t>(((int64_t)n_spokes + 3) / 4, 256 * 16);
    hipLaunchKernelGGL((owner_records<T, true>), dim3(pass_blocks), dim3(256), 0, s, rows, ctx, bias, (gran_t *)tagged, stride, n_spokes, a.k, ncs, vpl,
                       ncw, tag0);
    SgdArgs<T> args = a;
    args.owner_tag0 = tag0;
    gran_t *tg = (gran_t *)tagged;
    if (strict) n_team = 0;
    void *params[] = {&args, &recs, &own_off, &tg, &error, &n_owners, &n_team};
    const size_t lds = n_team > 0 ? kn.team_lds : 0;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(kn.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); // up to 131 KB (fp64, 6 mask words)
    hipError_t e = hipLaunchKernel(kn.fn, dim3((unsigned)(n_team + (n_owners - n_team + 3) / 4)), dim3(256), params, lds, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((owner_records<T, false>), dim3(pass_blocks), dim3(256), 0, s, rows, ctx, bias, (gran_t *)tagged, stride, n_spokes, a.k, ncs, vpl,
                       ncw, tag0);
    return hipGetLastError();
}
template hipError_t launch_owner_epoch<float>(const SgdArgs<float> &, int, bool, bool, const void *, const int64_t *, int, int, void *, int64_t, int,
                                              int *, uint32_t, hipStream_t);
template hipError_t launch_owner_epoch<double>(const SgdArgs<double> &, int, bool, bool, const void *, const int64_t *, int, int, void *, int64_t, int,
                                               int *, uint32_t, hipStream_t);

} // namespace cmi

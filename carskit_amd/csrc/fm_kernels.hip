// fm_kernels.hip -- gfx950 kernels for the reference's FM recommender
// (src/carskit/alg/cars/adaptation/dependent/FM.java:115-220): an ALS / coordinate-descent sweep, NOT SGD.
//
// The reference walks all p = numUsers+numItems+numConditions coordinates one after another and, for each,
// loops over ALL ratings with a dense feature vector (O(size*p*k) per sweep).  Every rating has at most three
// non-zero features -- its user (value 1), its item (1) and, if its context-combination id c is < numConditions
// (the reference's index quirk, FM.java:81-86), feature numUsers+numItems+c with value 1/numContextDims.
// Rows whose feature l is zero contribute exactly 0 to the numerator and exactly `reg` to the denominator of
// coordinate l and are not touched by its error update.  So
//   * a coordinate only needs the ratings in its support;
//   * the coordinates of one FIELD (all users / all items / all context features) have pairwise disjoint
//     supports, so their sequential updates commute exactly and run in parallel;
//   * the denominator is sum_{support} h^2 + size*reg.
// One sweep = 1 (w0) + 3 (w: users, items, contexts) + 3k (V, per factor) phases.
//
// What bounds a phase is HBM / fabric traffic per rating, so a sweep WRITES NO PER-RATING DATA AT ALL (round 4):
//   * errors[i] (FM.java:133-136, rewritten by every coordinate update :165,188,208) is not stored.  Each update adds
//     delta_l * x_il to the errors of its support, so  errors[i] = err0[i] + d0 + D[user] + D[item] + xc*D[ctx feature]
//     with err0 from cmi_fm_init and D[l] the running sum of coordinate l's deltas -- one fp64 per COORDINATE, kept beside
//     the column entry in tab[l] = {V[l][f], D[l]} (16 bytes, one gather serves both).  Equal to the reference's errors up
//     to the association of the additions (1e-16 relative; the tests hold the model to 1e-8 against the dense oracle).
//   * the reference's cache Q[i][f] = sum_l V[l][f] x_il (FM.java:134-146, 209-210) is not stored either: with three
//     features per rating it is V[u][f] + V[item][f] + xc*V[ctx][f], the same gathers.
//   * every field STREAMS its own copy of the ratings: nothing is gathered from per-rating arrays (round 3 did: 68 bytes fetched per
//     16 useful in the item phase).
//   * the only gathers left are the OTHER field's table entries, one 16-byte gather per rating and phase, L2 hits by construction.
//     Round 4 streamed 16-byte records sorted by (slice of the other id, own coordinate), a wave per 512 records: 160 us per launch,
//     bound by the L2 REQUEST rate -- every lane's gather in a 128-byte line of its own (tools/micro/gather16.hip: 210 G such
//     gathers/s on this part = 119 us for 25 M, whatever the slice size; the TCP sends one request for the lanes of an instruction
//     that fall into one line: two lanes per line 57 us, four 36 us).  Round 5 (the CELL stream, fm_kernels.hpp): the user and the
//     item field evaluate their records in GATHERED-ID order inside groups of ~4 900 coordinates (3 lanes per line at BASELINE C4's
//     share) and add a record's products to its coordinate's sums, which live in the workgroup for the whole group: parked in LDS at
//     the record's position in coordinate order and added left to right into register accumulators (fm_cell_kernel, the DEFAULT since
//     round 6: a fixed order, bit-reproducible like the reference's sweep; 135 us per launch, 18.2-18.6 ms per sweep), or -- the opt-in
//     CMI_FM_FLAG_RELAXED_SUMS -- in LDS, added with workgroup-scope fp64 atomics as the records are evaluated (fm_cell_atomic_kernel:
//     116-119 us, 15.8-16.3 ms; the order of a slot's additions varies run to run).  12-byte records, no per-piece partial sums through
//     memory, the coordinate update in the same launch (or one small kernel where a coordinate's sums come from several workgroups).
//   * the context field (a few records in a thousand): 16-byte records sorted by feature, one wave per feature, reduce + update in one
//     launch (fm_ctx_kernel).
// The default form adds every sum in a fixed order that depends on the data layout alone; the relaxed form's LDS atomics add a slot's
// records in whatever order its waves arrive (equal to rounding).  fp64 throughout; gather / stream work: no MFMA.
//
// reduce (-> [num | den] per coordinate) and apply are separable, so a multi-GPU host can all-reduce (num, den) between them; the
// fused sweep runs the identical arithmetic.
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

__device__ __forceinline__ int64_t fm_base(const FmArgs &a, int field) {
    return field == 0 ? 0 : (field == 1 ? (int64_t)a.n_users : (int64_t)a.n_users + a.n_items);
}

// The coordinate's current value: the linear weight, or column f of V out of the sweeps' working copy Vt (NOT tab[].x: that holds
// the column the GATHERS of the coming phases need, see fm_update).
__device__ __forceinline__ double fm_theta(const FmArgs &a, int f, int64_t bl) {
    return f < 0 ? a.w[bl] : a.Vt[(size_t)f * (size_t)((int64_t)a.n_users + a.n_items + a.n_conds) + (size_t)bl];
}

__device__ __forceinline__ FmRec fm_load_rec(const FmRec *rec, int64_t i) {
    // streamed once per phase: non-temporal, so the table slice keeps its place in L2
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d v = __builtin_nontemporal_load((const v2d *)rec + i);
    FmRec r;
    r.err0 = v.x;
    r.a = __double2loint(v.y);
    r.c = __double2hiint(v.y);
    return r;
}


// One record of FIELD's stream -> e' = its error without the own coordinate's D (the owner adds it) and
// h = x_il * (sum of the OTHER features' column entries times their values) = x*Q[i][f] - x*x*theta of FM.java:178,198
// (h = x for the linear weights).  One 16-byte gather per table touched.  N records at a time: all main gathers are
// issued before anything waits on one, and the context gathers sit behind ONE wave-uniform branch (context features
// are rare: only combination ids < numConditions have one), so the common case is one gather per record.
template <int FIELD, int N>
__device__ __forceinline__ void fm_rec_eval(const FmArgs &a, int f, const FmRec (&r)[N], double d0, double (&ep)[N], double (&h)[N]) {
    const int64_t jbase = a.n_users, cbase = (int64_t)a.n_users + a.n_items;
    if (FIELD == 2) {
        double2 tu[N], tj[N];
#pragma unroll
        for (int q = 0; q < N; ++q) {
            tu[q] = a.tab[r[q].a];
            tj[q] = a.tab[jbase + r[q].c];
        }
#pragma unroll
        for (int q = 0; q < N; ++q) {
            ep[q] = ((r[q].err0 + d0) + tu[q].y) + tj[q].y;
            // (the user's and the item's column entries out of Vt: tab[].x of the items already holds the next factor's column)
            h[q] = f < 0 ? a.xc : a.xc * (fm_theta(a, f, r[q].a) + fm_theta(a, f, jbase + r[q].c));
        }
    } else {
        double2 t[N];
        bool any_c = false;
#pragma unroll
        for (int q = 0; q < N; ++q) {
            t[q] = a.tab[(FIELD == 0 ? jbase : 0) + r[q].a];
            any_c |= r[q].c < a.n_conds;
        }
        double other[N];
#pragma unroll
        for (int q = 0; q < N; ++q) {
            ep[q] = (r[q].err0 + d0) + t[q].y;
            other[q] = t[q].x;
        }
        if (__builtin_amdgcn_ballot_w64(any_c) != 0) {
#pragma unroll
            for (int q = 0; q < N; ++q) {
                const bool has_c = r[q].c < a.n_conds;
                const double2 tc = a.tab[cbase + (has_c ? r[q].c : 0)]; // tab has p + 1 entries: in bounds even with no conditions
                if (has_c) {
                    ep[q] += a.xc * tc.y;
                    other[q] += a.xc * tc.x;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < N; ++q) h[q] = f < 0 ? 1.0 : other[q];
    }
}

// The update (FM.java:181-190, 201-211): theta' = -num / (den + size*reg); the errors of the support move by (theta' - theta) x_il,
// which is folded into the coordinate's running delta D = tab[].y.  tab[].x is left holding column a.xcol: a user update leaves the
// column it just wrote (the item phase of the same factor gathers it), an item / context update leaves the NEXT factor's column (the
// next reader of those entries is the next factor's user phase) -- so a sweep never needs a pass that reloads the columns
// (round 4's fm_col_load per factor: 9.5 us of a 290-us factor).
__device__ __forceinline__ void fm_update(const FmArgs &a, int f, int64_t bl, double2 t, double theta, double num, double den) {
    const double reg = f < 0 ? a.regLw : a.regLf;
    const double upd = 0.0 - num / (den + (double)a.global_size * reg);
    const double delta = upd - theta;
    const size_t p = (size_t)((int64_t)a.n_users + a.n_items + a.n_conds);
    if (f < 0) a.w[bl] = upd;
    else a.Vt[(size_t)f * p + (size_t)bl] = upd;
    const double x = a.xcol < 0 ? t.x : (a.xcol == f ? upd : a.Vt[(size_t)a.xcol * p + (size_t)bl]);
    a.tab[bl] = make_double2(x, t.y + delta);
}

// ---- fields 0 / 1: the cell stream (fm_kernels.hpp) -------------------------------------------------------------------------------
// One coordinate's result from its sums A = sum e'h, B = sum h, C = sum h^2 over its support (e' = the error without the coordinate's
// own running delta D, h as in fm_rec_eval):  num = sum (e' + D - theta h) h = (A + D B) - theta C,  den = C   (FM.java:178-184, 198-204).
// fused: the update itself (FM.java:181-190, 201-211): theta' = -num / (den + size*reg), D += theta' - theta; else [num | den] -> part.
template <int FIELD>
__device__ __forceinline__ void fm_coord_out(const FmArgs &a, int f, int l, double A, double B, double C, bool fused) {
    const int64_t base = fm_base(a, FIELD);
    const double2 t = a.tab[base + l];
    const double theta = fm_theta(a, f, base + l);
    const double num = (A + t.y * B) - theta * C, den = C;
    if (!fused) {
        a.part[l] = num;
        a.part[a.ord[FIELD].count + l] = den;
        return;
    }
    fm_update(a, f, base + l, t, theta, num, den);
}

constexpr int FMC_HQ = FMC_RCAP / FMC_THREADS / 2;                   // records per thread and HALF batch
constexpr int FMC_NSL = (FMC_SLOTS + FMC_THREADS - 1) / FMC_THREADS; // slots per thread
static_assert(FMC_RCAP % (2 * FMC_THREADS) == 0 && FMC_RCAP <= 16384 && FMC_RCAP * 16 <= 160 * 1024, "batch = whole rounds of the workgroup, 14-bit positions, LDS");

// Half a batch's registers in flight: the records (8-byte error, packed word)
struct FmcHalf {
    double e0[FMC_HQ];
    uint32_t pk[FMC_HQ];
};

// W0: the w0 phase (FM.java:153-158): per slot sum(err_i - w0) -> w0part.  FUSED: coordinates with ONE slot are updated by this launch.
//
// Software pipeline of a workgroup (one per CU: the parking area is most of the LDS), in HALF batches (half = rounds [0, HQ) / [HQ, 2 HQ) of
// the workgroup over the batch's records), register sets X (first halves) and Y (second halves):
//     park X(i)        <- waits for its gathers            |  a thread keeps <= 2 half batches of records and one of gathered entries:
//     gathers Y(i)     <- waits for its records            |  what the register file holds at 16 waves per CU
//     request X(i+1)
//     park Y(i)
//     barrier
//     gathers X(i+1), request Y(i+1)
//     add the runs of batch i out of LDS into the threads' REGISTER accumulators (a thread owns slots t, t + THREADS, ... of the block)
//     barrier
// so half a batch of gathers AND half a batch of record requests are in flight while the LDS phase and the barriers run.  The pipelined
// batches are straight-line code: indices past a batch's end are clamped for the loads and parked at positions no record has.  Ratings with a
// context feature (FM.java:81-86: rare, only combination ids < numConditions have one) are kept out of them: they form the block's last
// batches (`flag0`), which carry their full ids in side arrays and are walked without the pipeline.
template <int FIELD, bool W0, bool FUSED>
__global__ __launch_bounds__(FMC_THREADS) void fm_cell_kernel(FmArgs a, int f) {
    __shared__ double2 park[FMC_RCAP];
    const FmCells &c = a.cell[FIELD];
    const int b = blockIdx.x;
    const unsigned t = threadIdx.x;
    const int s0 = c.slot_off[b], ns = c.slot_off[b + 1] - s0;
    double A[FMC_NSL], B[FMC_NSL], C[FMC_NSL];
#pragma unroll
    for (int q = 0; q < FMC_NSL; ++q) A[q] = B[q] = C[q] = 0.0;
    const double d0 = *a.d0;
    const int b0 = c.bat_off[b], bf = c.flag0[b], be = c.bat_off[b + 1];
    FmcHalf X, Y;
    double2 tt[FMC_HQ];
    uint32_t po[FMC_NSL];
    const FmBatch none = FmBatch{0, 0, 0, 0, 0, 0};

    // records of rounds [half * HQ, half * HQ + HQ) of batch d
    auto load = [&](const FmBatch &d, int half, FmcHalf &r) {
        const double *eb = c.err0 + d.rec0;
        const uint32_t *pb = c.pk + d.rec0;
        const unsigned last = d.n > 0 ? (unsigned)d.n - 1u : 0u;
#pragma unroll
        for (int q = 0; q < FMC_HQ; ++q) {
            unsigned i = (unsigned)(half * FMC_HQ + q) * FMC_THREADS + t;
            i = i < last ? i : last; // clamped: no branch splits the loads
            r.e0[q] = __builtin_nontemporal_load(eb + i);
            r.pk[q] = __builtin_nontemporal_load(pb + i);
        }
    };
    auto gather = [&](const FmBatch &d, const FmcHalf &r) {
        const double2 *tb = a.tab + d.tab0;
#pragma unroll
        for (int q = 0; q < FMC_HQ; ++q) tt[q] = tb[r.pk[q] & 0x1FFFFu];
    };
    auto eval_park = [&](const FmBatch &d, int half, const FmcHalf &r) {
#pragma unroll
        for (int q = 0; q < FMC_HQ; ++q) {
            const unsigned i = (unsigned)(half * FMC_HQ + q) * FMC_THREADS + t;
            const unsigned pos = i < (unsigned)d.n ? (r.pk[q] >> 17) & 0x3FFFu : i; // past the end: its own index, a position no record of the batch has
            park[pos] = make_double2((r.e0[q] + d0) + tt[q].y, f < 0 ? 1.0 : tt[q].x);
        }
    };
    // the thread's slot boundaries of batch d, two 16-bit positions per slot in one (unaligned) 32-bit load
    auto load_po = [&](const FmBatch &d) {
        const uint16_t *pb = c.poff + d.poff0;
#pragma unroll
        for (int q = 0; q < FMC_NSL; ++q) {
            const unsigned s = (unsigned)q * FMC_THREADS + t;
            typedef uint32_t __attribute__((aligned(2))) u32_a2;
            po[q] = *(const u32_a2 *)(pb + (s < (unsigned)ns ? s : 0u));
        }
    };
    auto add_runs = [&]() {
        // the thread's slots side by side: FMC_NSL independent LDS reads per step instead of one dependent chain per slot
        int o[FMC_NSL], e[FMC_NSL], longest = 0;
#pragma unroll
        for (int q = 0; q < FMC_NSL; ++q) {
            o[q] = (int)(po[q] & 0xFFFFu);
            e[q] = (unsigned)q * FMC_THREADS + t < (unsigned)ns ? (int)(po[q] >> 16) : o[q];
            longest = e[q] - o[q] > longest ? e[q] - o[q] : longest;
        }
        for (int step = 0; step < longest; ++step) {
#pragma unroll
            for (int q = 0; q < FMC_NSL; ++q)
                if (o[q] + step < e[q]) {
                    const double2 v = park[o[q] + step];
                    A[q] += v.x * v.y;
                    B[q] += v.y;
                    C[q] += v.y * v.y;
                }
        }
    };

    FmBatch cur = b0 < bf ? c.bat[b0] : none;
    load(cur, 0, X);
    gather(cur, X);
    load(cur, 1, Y);
    for (int bi = b0; bi < bf; ++bi) {
        const FmBatch nxt = bi + 1 < bf ? c.bat[bi + 1] : none;
        load_po(cur);
        eval_park(cur, 0, X);
        gather(cur, Y);
        load(nxt, 0, X);
        eval_park(cur, 1, Y);
        __syncthreads();
        gather(nxt, X);
        load(nxt, 1, Y);
        add_runs();
        __syncthreads();
        cur = nxt;
    }
    // the block's ratings with a context feature: full ids from the side arrays, no pipeline
    const int64_t obase = fm_base(a, 1 - FIELD), cbase = (int64_t)a.n_users + a.n_items;
    for (int bi = bf; bi < be; ++bi) {
        const FmBatch d = c.bat[bi];
        load_po(d);
        for (unsigned i = t; i < (unsigned)d.n; i += FMC_THREADS) {
            const double e0 = c.err0[(int64_t)d.rec0 + i];
            const uint32_t pk = c.pk[(int64_t)d.rec0 + i];
            const double2 to = a.tab[obase + c.fo[(int64_t)d.tab0 + i]], tc = a.tab[cbase + c.fcx[(int64_t)d.tab0 + i]];
            park[(pk >> 17) & 0x3FFFu] = make_double2(((e0 + d0) + to.y) + a.xc * tc.y, f < 0 ? 1.0 : to.x + a.xc * tc.x);
        }
        __syncthreads();
        add_runs();
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < FMC_NSL; ++q) {
        const int s = q * FMC_THREADS + (int)t;
        if (s >= ns) continue;
        const int g = s0 + s, lc = c.slot_coord[g], l = lc & 0x7FFFFFFF;
        if (W0) {
            const double dl = a.tab[fm_base(a, FIELD) + l].y;
            c.w0part[g] = (A[q] + dl * B[q]) - *a.w0 * B[q];
        } else if (lc < 0) {
            c.partial3[3 * (int64_t)g] = A[q];
            c.partial3[3 * (int64_t)g + 1] = B[q];
            c.partial3[3 * (int64_t)g + 2] = C[q];
        } else {
            fm_coord_out<FIELD>(a, f, l, A[q], B[q], C[q], FUSED);
        }
    }
}

// ---- the relaxed form (a.atomic, CMI_FM_FLAG_RELAXED_SUMS; the default until round 6): the same cell stream without parking -- every wave walks its own slice of each batch (records
// requested one batch ahead) and adds the three products of a record straight into LDS accumulators with ds_add_f64 (the packed word's
// 14-bit field holds the record's SLOT).  No barriers between batches, no second pass over LDS, no slot boundaries to stream: the
// waves run independently, so stream, gathers, VALU and LDS overlap by themselves.  The order of the additions is not fixed: sums vary
// in their last bits run to run (which is why fm_cell_kernel above is the default).  116-119 us per launch against 135.
template <int FIELD, bool W0, bool FUSED>
__global__ __launch_bounds__(FMC_THREADS) void fm_cell_atomic_kernel(FmArgs a, int f) {
    __shared__ double acc[3][FMC_SLOTS];
    const FmCells &c = a.cell[FIELD];
    const int b = blockIdx.x;
    const unsigned t = threadIdx.x, lane = t & 63;
    const int s0 = c.slot_off[b], ns = c.slot_off[b + 1] - s0;
    for (int s = (int)t; s < FMC_SLOTS; s += FMC_THREADS) acc[0][s] = acc[1][s] = acc[2][s] = 0.0;
    __syncthreads();
    const double d0 = *a.d0;
    const int b0 = c.bat_off[b], bf = c.flag0[b], be = c.bat_off[b + 1];
    constexpr int PW = FMC_CHUNK / 64; // records per lane and batch: in this form a batch is a CHUNK of <= FMC_CHUNK records, a wave's unit of work
    const FmBatch none = FmBatch{0, 0, 0, 0, 0, 0};
    // wave w walks chunks b0 + w, b0 + w + 16, ... and never waits for another wave before the epilogue.  Three stages in flight per
    // wave: the records of step s + 2 and the gathers of step s + 1 are outstanding while step s adds into LDS (the memory pipe and the
    // LDS pipe of a CU are busy ~59 and ~48 us of a launch: with a wave's stages back to back, and the 16 waves in phase, they added up).
    const int wv = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    const int nst = bf - b0 - wv > 0 ? (bf - b0 - wv + 15) / 16 : 0;
    auto desc = [&](int st) -> FmBatch { return st < nst ? c.bat[b0 + wv + 16 * st] : none; };
    double e0[3][PW];
    uint32_t pk[3][PW];
    double2 tt[2][PW];
    auto load = [&](const FmBatch &d, int buf) {
        const double *eb = c.err0 + d.rec0;
        const uint32_t *pb = c.pk + d.rec0;
        const unsigned last = d.n > 0 ? (unsigned)d.n - 1u : 0u;
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            unsigned i = q * 64 + lane;
            i = i < last ? i : last;
            e0[buf][q] = __builtin_nontemporal_load(eb + i);
            pk[buf][q] = __builtin_nontemporal_load(pb + i);
        }
    };
    auto gather = [&](const FmBatch &d, int buf, int tb) {
        const double2 *base = a.tab + d.tab0;
#pragma unroll
        for (int q = 0; q < PW; ++q) tt[tb][q] = base[pk[buf][q] & 0x1FFFFu];
    };
    auto add = [&](const FmBatch &d, int buf, int tb) {
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const unsigned i = q * 64 + lane;
            if (i < (unsigned)d.n) {
                const double ep = (e0[buf][q] + d0) + tt[tb][q].y, hh = f < 0 ? 1.0 : tt[tb][q].x;
                const unsigned sl = (pk[buf][q] >> 17) & 0x3FFFu;
                __hip_atomic_fetch_add(&acc[0][sl], ep * hh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&acc[1][sl], hh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&acc[2][sl], hh * hh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };
    FmBatch dA = desc(0), dB = desc(1), dC = desc(2), dD = desc(3); // steps s, s + 1, s + 2, s + 3
    load(dA, 0);
    load(dB, 1);
    gather(dA, 0, 0);
    for (int st = 0; st < nst; st += 6) {
#pragma unroll
        for (int ph = 0; ph < 6; ++ph) {
            if (st + ph >= nst) break;
            load(dC, (ph + 2) % 3);
            gather(dB, (ph + 1) % 3, (ph + 1) & 1);
            add(dA, ph % 3, ph & 1);
            dA = dB;
            dB = dC;
            dC = dD;
            dD = desc(st + ph + 4);
        }
    }
    const int64_t obase = fm_base(a, 1 - FIELD), cbase = (int64_t)a.n_users + a.n_items;
    for (int bi = bf; bi < be; ++bi) { // ratings with a context feature
        const FmBatch d = c.bat[bi];
        for (unsigned r = t; r < (unsigned)d.n; r += FMC_THREADS) {
            const double e = c.err0[(int64_t)d.rec0 + r];
            const uint32_t w = c.pk[(int64_t)d.rec0 + r];
            const double2 to = a.tab[obase + c.fo[(int64_t)d.tab0 + r]], tc = a.tab[cbase + c.fcx[(int64_t)d.tab0 + r]];
            const double ep = ((e + d0) + to.y) + a.xc * tc.y, hh = f < 0 ? 1.0 : to.x + a.xc * tc.x;
            const unsigned sl = (w >> 17) & 0x3FFFu;
            __hip_atomic_fetch_add(&acc[0][sl], ep * hh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&acc[1][sl], hh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&acc[2][sl], hh * hh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    for (int s = (int)t; s < ns; s += FMC_THREADS) {
        const int g = s0 + s, lc = c.slot_coord[g], l = lc & 0x7FFFFFFF;
        const double A = acc[0][s], B = acc[1][s], C = acc[2][s];
        if (W0) {
            const double dl = a.tab[fm_base(a, FIELD) + l].y;
            c.w0part[g] = (A + dl * B) - *a.w0 * B;
        } else if (lc < 0) {
            c.partial3[3 * (int64_t)g] = A;
            c.partial3[3 * (int64_t)g + 1] = B;
            c.partial3[3 * (int64_t)g + 2] = C;
        } else {
            fm_coord_out<FIELD>(a, f, l, A, B, C, FUSED);
        }
    }
}

// complex coordinates (several slots: a hot run spread over slots, a block's id-range parts, a giant's blocks): their slots' sums in
// slot order, then the same output.  cplx[4 i ..] = coordinate, first slot, parts, stride between parts, slots per part is cplx_vs.
template <int FIELD>
__global__ __launch_bounds__(256) void fm_cplx_kernel(FmArgs a, int f, int fused) {
    const FmCells &c = a.cell[FIELD];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= c.n_cplx) return;
    const int4 d = ((const int4 *)c.cplx)[i];
    const int l = d.x, g0 = d.y, parts = d.z & 0xFFFF, vs = d.z >> 16, stride = d.w;
    double A = 0.0, B = 0.0, C = 0.0;
    for (int h = 0; h < parts; ++h)
        for (int v = 0; v < vs; ++v) {
            const int64_t g = (int64_t)g0 + (int64_t)h * stride + v;
            A += c.partial3[3 * g];
            B += c.partial3[3 * g + 1];
            C += c.partial3[3 * g + 2];
        }
    fm_coord_out<FIELD>(a, f, l, A, B, C, fused != 0);
}

// part ([num | den] of the phase's field, all-reduced by a multi-GPU host) -> the coordinate update
template <int FIELD>
__global__ __launch_bounds__(256) void fm_apply_kernel(FmArgs a, int f) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int count = a.ord[FIELD].count;
    if (l >= count) return;
    const int64_t base = fm_base(a, FIELD);
    fm_update(a, f, base + l, a.tab[base + l], fm_theta(a, f, base + l), a.part[l], a.part[count + l]);
}

// The context field's phase: one 256-thread workgroup per context feature adds its piece (thread-strided sums, a fixed butterfly per wave, the waves in order)
// and -- fused -- applies the update; else [num | den] -> part.  A few ratings in a thousand have such a feature (FM.java:81-86), so
// this is a latency-bound launch of n_conds waves: one kernel instead of round 4's chunked reduce + finish pair (12 + 5 us -> 6 us).
__global__ __launch_bounds__(256) void fm_ctx_kernel(FmArgs a, int f, int fused) {
    __shared__ double red[2][4];
    const FmOrder &o = a.ord[2];
    const int l = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = fm_base(a, 2);
    const double2 t = a.tab[base + l];
    const double theta = fm_theta(a, f, base + l), dl = a.xc * t.y, d0 = *a.d0; // the support's errors moved by delta * x_il
    double num = 0.0, den = 0.0;
    for (int i = o.piece_off[l] + tid; i < o.piece_off[l + 1]; i += 256) { // thread-strided sums in record order
        FmRec rr[1] = {fm_load_rec(o.rec, i)};
        double ep[1], hh[1];
        fm_rec_eval<2>(a, f, rr, d0, ep, hh);
        const double et = ep[0] + dl;
        num += (et - theta * hh[0]) * hh[0];
        den += hh[0] * hh[0];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { // a fixed butterfly per wave, then the four waves in order
        num += __shfl_xor(num, m, 64);
        den += __shfl_xor(den, m, 64);
    }
    if (lane == 0) {
        red[0][wave] = num;
        red[1][wave] = den;
    }
    __syncthreads();
    if (tid != 0) return;
    num = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    den = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    if (fused) fm_update(a, f, base + l, t, theta, num, den);
    else {
        a.part[l] = num;
        a.part[o.count + l] = den;
    }
}

// tab[l].x = Vt[f][l] for entries [lo, hi): only when phases are driven out of the sweep's order (fm_update leaves the right columns)
__global__ __launch_bounds__(256) void fm_col_load(FmArgs a, int f, int64_t lo, int64_t hi, int64_t p) {
    for (int64_t l = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; l < hi; l += (int64_t)gridDim.x * 256)
        a.tab[l].x = a.Vt[(size_t)f * (size_t)p + (size_t)l]; // .y (the coordinate's running delta sum) is kept
}

// dst[c][r] = src[r][c] (V <-> Vt), 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void fm_transpose_kernel(const double *src, double *dst, int64_t rows, int64_t cols) {
    __shared__ double tile[32][33];
    const int64_t r0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = src[(size_t)(r0 + i) * cols + (size_t)(c0 + tx)];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) dst[(size_t)(c0 + i) * rows + (size_t)(r0 + tx)] = tile[tx][i];
}

// w0 phase, reduce: part[0] = sum over the pieces' sums of (err_i - w0) (fixed two-stage tree)
__global__ __launch_bounds__(256) void fm_w0_reduce1(FmArgs a, double *scratch) {
    __shared__ double lds[6];
    const FmCells &o = a.cell[0];
    const int64_t slots = o.n_slots;
    const int64_t chunk = (slots + gridDim.x - 1) / gridDim.x;
    const int64_t b = (int64_t)blockIdx.x * chunk, e = (b + chunk) < slots ? (b + chunk) : slots;
    double s = 0.0;
    for (int64_t i = b + threadIdx.x; i < e; i += 256) s += o.w0part[i];
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
// w0 phase, apply: w0' = -part[0]/(size + regLw); every error moves by w0' - w0 (FM.java:153-169): d0 takes it
__global__ void fm_w0_apply(FmArgs a) {
    const double w0 = *a.w0;
    // `size + regLw` is int + float in the reference (FM.java:47,161): Java's binary numeric promotion makes it a FLOAT sum (at C4's
    // 25 M ratings regLw vanishes in it) -- found by executing the reference's source, tests/test_reference_src_golden.py
    const double upd = 0.0 - a.part[0] / (double)((float)(int)a.global_size + (float)a.regLw);
    *a.d0 = *a.d0 + (upd - w0);
    *a.w0 = upd;
}

// pre-pass (FM.java:117-146): E[i] = err0[i] = r_i - predict(i) per rating, in the caller's order (Q is not materialised, see the
// header); the running delta sums start at zero.  One wave per rating.
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
            a.E[i] = a.r[i] - (pred + 0.5 * pair);
        }
    }
}
// the three streams copy their err0 from E, so the three copies are the same numbers
__global__ __launch_bounds__(256) void fm_init_spread(FmArgs a) {
    FmRec *rc = const_cast<FmRec *>(a.ord[2].rec);
    const int64_t n2 = a.ord[2].n_rec, p = (int64_t)a.n_users + a.n_items + a.n_conds;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        a.cell[0].err0[i] = a.E[a.src[0][i]];
        a.cell[1].err0[i] = a.E[a.src[1][i]];
        if (i < n2) rc[i].err0 = a.E[a.src[2][i]];
        if (i < p) a.tab[i].y = 0.0;
    }
    for (int64_t i = a.n + (int64_t)blockIdx.x * 256 + threadIdx.x; i < p; i += (int64_t)gridDim.x * 256) a.tab[i].y = 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.d0 = 0.0;
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

static hipError_t launch_ctx(const FmArgs &a, int f, int fused, hipStream_t s) {
    if (a.ord[2].count <= 0) return hipSuccess;
    hipLaunchKernelGGL(fm_ctx_kernel, dim3(a.ord[2].count), dim3(256), 0, s, a, f, fused);
    return hipGetLastError();
}

template <int FIELD, bool W0, bool FUSED>
static hipError_t launch_cells(const FmArgs &a, int f, hipStream_t s) {
    const FmCells &c = a.cell[FIELD];
    if (c.n_blocks <= 0) return hipSuccess;
    if (a.atomic) hipLaunchKernelGGL((fm_cell_atomic_kernel<FIELD, W0, FUSED>), dim3(c.n_blocks), dim3(FMC_THREADS), 0, s, a, f);
    else hipLaunchKernelGGL((fm_cell_kernel<FIELD, W0, FUSED>), dim3(c.n_blocks), dim3(FMC_THREADS), 0, s, a, f);
    if (!W0 && c.n_cplx > 0) {
        if (FIELD == 0) hipLaunchKernelGGL(fm_cplx_kernel<0>, dim3((c.n_cplx + 255) / 256), dim3(256), 0, s, a, f, (int)FUSED);
        else hipLaunchKernelGGL(fm_cplx_kernel<1>, dim3((c.n_cplx + 255) / 256), dim3(256), 0, s, a, f, (int)FUSED);
    }
    return hipGetLastError();
}

hipError_t fm_launch_phase(const FmArgs &a, int field, int f, int mode, hipStream_t s) {
    const bool fused = mode == 2;
    switch (field) {
    case 0: return fused ? launch_cells<0, false, true>(a, f, s) : launch_cells<0, false, false>(a, f, s);
    case 1: return fused ? launch_cells<1, false, true>(a, f, s) : launch_cells<1, false, false>(a, f, s);
    default: return launch_ctx(a, f, fused ? 1 : 0, s);
    }
}

hipError_t fm_launch_reduce_only(const FmArgs &a, int field, int f, hipStream_t s) {
    switch (field) {
    case 0: return launch_cells<0, false, false>(a, f, s);
    case 1: return launch_cells<1, false, false>(a, f, s);
    default: return launch_ctx(a, f, 0, s);
    }
}

hipError_t fm_launch_apply(const FmArgs &a, int field, int f, hipStream_t s) {
    const int count = a.ord[field].count;
    if (count <= 0) return hipSuccess;
    const dim3 grid((count + 255) / 256), block(256);
    switch (field) {
    case 0: hipLaunchKernelGGL(fm_apply_kernel<0>, grid, block, 0, s, a, f); break;
    case 1: hipLaunchKernelGGL(fm_apply_kernel<1>, grid, block, 0, s, a, f); break;
    default: hipLaunchKernelGGL(fm_apply_kernel<2>, grid, block, 0, s, a, f); break;
    }
    return hipGetLastError();
}

hipError_t fm_launch_col_load(const FmArgs &a, int field, int f, hipStream_t s) {
    const int64_t p = (int64_t)a.n_users + a.n_items + a.n_conds;
    const int64_t lo = field == 0 ? 0 : field == 1 ? a.n_users : (int64_t)a.n_users + a.n_items;
    const int64_t hi = field == 0 ? a.n_users : field == 1 ? (int64_t)a.n_users + a.n_items : p;
    if (hi <= lo) return hipSuccess;
    int64_t blocks = (hi - lo + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fm_col_load, dim3((unsigned)blocks), dim3(256), 0, s, a, f, lo, hi, p);
    return hipGetLastError();
}

hipError_t fm_launch_transpose(const double *src, double *dst, int64_t rows, int64_t cols, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return hipSuccess;
    hipLaunchKernelGGL(fm_transpose_kernel, dim3((unsigned)((rows + 31) / 32), (unsigned)((cols + 31) / 32)), dim3(256), 0, s, src, dst,
                       rows, cols);
    return hipGetLastError();
}

hipError_t fm_launch_w0_reduce(const FmArgs &a, double *scratch, hipStream_t s) {
    if (hipError_t e = launch_cells<0, true, false>(a, -1, s)) return e;
    const int64_t slots = a.cell[0].n_slots;
    int nblk = (int)((slots + 65535) / 65536);
    if (nblk < 1) nblk = 1;
    if (nblk > 256) nblk = 256;
    hipLaunchKernelGGL(fm_w0_reduce1, dim3(nblk), dim3(256), 0, s, a, scratch);
    hipLaunchKernelGGL(fm_w0_reduce2, dim3(1), dim3(256), 0, s, a, scratch, nblk);
    return hipGetLastError();
}

hipError_t fm_launch_w0_apply(const FmArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(fm_w0_apply, dim3(1), dim3(1), 0, s, a);
    return hipGetLastError();
}

hipError_t fm_launch_init(const FmArgs &a, hipStream_t s) {
    const int64_t p = (int64_t)a.n_users + a.n_items + a.n_conds;
    if (a.n > 0) {
        int64_t blocks = (a.n + 3) / 4;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(fm_init_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    }
    int64_t blocks = ((a.n > p ? a.n : p) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fm_init_spread, dim3((unsigned)blocks), dim3(256), 0, s, a);
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

// camfc_pipe.hip -- CAMF_C (src/carskit/alg/cars/adaptation/dependent/dev/CAMF_C.java:79-131) as ONE software-pipelined wave.
//
// CAMF_C's shared condBias vector makes every tuple depend on the one before it: the epoch is a single dependency chain and the only
// lever is the latency of one link.  sgd_serial_fast (mf_sgd_kernels.hip) requests the rows of tuple t+1 while it computes tuple t, so
// every link still waits most of a memory round trip (0.62 us per tuple measured on DePaulMovie and on the Frappe shape -- one CPU core
// does a k = 64 link in 0.095 us).  This kernel requests the rows of tuple t+D (D = 8 for one-instruction rows) while it computes
// tuple t, so the round trip is hidden behind D links and a link costs what its arithmetic chain costs:
//   * lane l owns the MAXC consecutive factors [l*MAXC, (l+1)*MAXC) of P[u] and Q[j] -- one 4/8/16-byte load or store per row;
//   * a ring of D register slots receives the requested rows; the loop body is unrolled D times so every slot is a fixed register;
//   * a requested row is STALE when one of the D tuples processed since the request wrote it.  The ids of the last D tuples sit in
//     one VGPR (lane s = ring slot s): one v_cmp + ballot per side says whether any of them is this tuple's user / item -- if so the
//     freshest copy is taken from an LDS ring that every tuple writes its updated rows to (the previous tuple's rows are simply
//     still in registers: consecutive CRS tuples share their user).  Writers further back than D tuples had issued their stores before
//     the request; a wave's stores and later loads of one address stay ordered in the vector-memory pipeline;
//   * condBias lives in a register (lane c owns condBias[c], n_conds <= 64): reading an entry is a v_readlane, no LDS round trip;
//   * a tuple's condition ids travel as one packed 64-bit word per tuple (<= 8 dimensions, 8 bits each), staged 64 tuples at a time
//     in registers, the next chunk's ids requested a chunk ahead;
//   * no memory operation sits under a branch in the steady state, so the compiler counts outstanding requests exactly
//     (s_waitcnt vmcnt(N), never a drain) -- the lesson of the hub-chain kernel.
// Arithmetic per element is the expression every other non-strict kernel uses; the dot is a DPP tree sum, the deviations are added
// one by one in condition order like the reference.  The last n mod 64 tuples take a plain loop (requests after stores).
#include "mf_sgd_kernels.hpp"
#include "sgd_device.hpp"

#include <cstdlib>

namespace cmi {
namespace {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float pdpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double pdpp(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int prl(int x, int lane) { return __builtin_amdgcn_readlane(x, lane); }
__device__ __forceinline__ float prl(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
__device__ __forceinline__ double prl(double x, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
__device__ __forceinline__ unsigned long long prl(unsigned long long x, int lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}
// lane `slot` (a constant after unrolling) of x := v (wave-uniform): one v_cndmask under a loop-invariant lane mask
template <typename V>
__device__ __forceinline__ V pwl(V x, V v, int slot, int lane) { return lane == slot ? v : x; }
template <typename T>
__device__ __forceinline__ T pwave_sum(T x) { // fixed tree, result uniform
    x += pdpp<0x128, 0xf>(x);
    x += pdpp<0x124, 0xf>(x);
    x += pdpp<0x122, 0xf>(x);
    x += pdpp<0x121, 0xf>(x);
    x += pdpp<0x142, 0xa>(x);
    x += pdpp<0x143, 0xc>(x);
    return prl(x, 63);
}

template <typename T, int MAXC>
struct alignas(sizeof(T) * MAXC) RowVec { T v[MAXC]; };

// FULL: k == 64 * MAXC, one aligned vector access per row; otherwise element-wise with a mask (k < 64 * MAXC)
template <typename T, int MAXC, bool FULL>
__device__ __forceinline__ RowVec<T, MAXC> load_row(const T *tab, int row, int k, int lane) {
    RowVec<T, MAXC> r;
    if (FULL) {
        r = *reinterpret_cast<const RowVec<T, MAXC> *>(tab + (size_t)row * k + (size_t)lane * MAXC);
    } else {
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int f = lane * MAXC + c;
            r.v[c] = tab[(size_t)row * k + (f < k ? f : 0)];
            if (f >= k) r.v[c] = (T)0;
        }
    }
    return r;
}
template <typename T, int MAXC, bool FULL>
__device__ __forceinline__ void store_row(T *tab, int row, int k, int lane, const RowVec<T, MAXC> &r) {
    if (FULL) {
        *reinterpret_cast<RowVec<T, MAXC> *>(tab + (size_t)row * k + (size_t)lane * MAXC) = r;
    } else {
        // a store under a lane mask is a memory operation under control flow, and the compiler then stops counting outstanding
        // requests exactly; so the lanes past k store too -- the row's LAST element, with the value its owner stores (same word, same
        // value: harmless, like the all-lane bias stores)
        const int lo = (k - 1) / MAXC, co = (k - 1) % MAXC;
        T last = prl(r.v[0], lo);
#pragma unroll
        for (int c = 1; c < MAXC; ++c)
            if (co == c) last = prl(r.v[c], lo);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int f = lane * MAXC + c;
            tab[(size_t)row * k + (f < k ? f : k - 1)] = f < k ? r.v[c] : last;
        }
    }
}

template <typename T>
struct PipeConst {
    T lr, regU, regI, regB, regC, gm;
};

// One CAMF_C update on rows held in registers.  On ONE wave a link costs what its dependent instructions cost -- about 8-10 cycles per
// dependent VALU operation, 20 per v_readlane and per taken branch (tools/micro/one_wave_clock.hip) -- so DM, the number of context
// dimensions, is a template parameter (no inner loop; an absent condition, 0xff, contributes an exact +0) and the tuple's uniform loss
// terms are returned as ONE float that the caller parks in a lane; they are summed in double once per 64 tuples.
// BCL: condBias lives in LDS (s_bc, more than 64 conditions -- the Frappe file has 343) instead of in lane c of one register.  The tuple's
// condition ids then travel as 16-bit fields of TWO packed words (pc: dimensions 0-3, pc2: 4-7; 0xffff = absent); lane d < DM reads and
// later rewrites s_bc[its condition] -- one ds_read / ds_write for all dimensions, ordered after the previous tuple's write by the wave's
// in-order LDS queue -- and the values reach the uniform sum through v_readlane, added one by one like in the register form.
template <typename T, int MAXC, int DM, bool BCL>
__device__ __forceinline__ T camfc_step_dm(RowVec<T, MAXC> &p, RowVec<T, MAXC> &q, T &bu, T &bj, T &bcreg, T *s_bc, const T rr,
                                           const unsigned long long pc, const unsigned long long pc2, const int lane,
                                           const PipeConst<T> &h, double &acc_reg, double &acc_ctx) {
    T part = 0;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) part += p.v[c] * q.v[c];
    const T dot = pwave_sum(part);
    T pred = h.gm;
    pred += bu;
    pred += bj;
    pred += dot;
    const unsigned lo = (unsigned)pc, hi = (unsigned)(pc >> 32);
    bool mine = false;
    unsigned mycond = 0;
    if (BCL) {
        const unsigned lo2 = (unsigned)pc2, hi2 = (unsigned)(pc2 >> 32);
        // lane d's own condition: 16-bit field d of (pc, pc2)
        const unsigned w = lane < 2 ? lo : lane < 4 ? hi : lane < 6 ? lo2 : hi2;
        mycond = (w >> (16 * (lane & 1))) & 0xffffu;
        mine = lane < DM && mycond != 0xffffu;
        bcreg = mine ? s_bc[mycond] : (T)0; // here bcreg is a per-tuple temporary: lane d holds the d-th deviation
#pragma unroll
        for (int d = 0; d < DM; ++d) pred += prl(bcreg, d); // absent dimensions contribute an exact +0
    } else {
#pragma unroll
        for (int d = 0; d < DM; ++d) { // the reference adds the deviations one by one, in condition order (CAMF_C.java:98-101)
            const unsigned cond = (d < 4 ? lo >> (8 * d) : hi >> (8 * (d - 4))) & 0xffu;
            const bool present = cond != 0xffu;
            const T got = prl(bcreg, (int)(cond & 63u));
            pred += present ? got : (T)0;
            mine = mine || (present && lane == (int)cond);
        }
    }
    const T decay = h.regC * bcreg;
    const T e = rr - pred;
    T l = e * e;
    {
        const T nb = bu + h.lr * (e - h.regB * bu);
        l += (h.regB * bu) * bu;
        bu = nb;
    }
    {
        const T nb = bj + h.lr * (e - h.regB * bj);
        l += (h.regB * bj) * bj;
        bj = nb;
    }
    if (mine) {
        acc_ctx += (double)bcreg; // plain sum, weighted by regB at the end (reference quirk, CAMF_C.java:110,115)
        bcreg = bcreg + h.lr * (e - decay);
        if (BCL) s_bc[mycond] = bcreg;
    }
    T reg_part = 0;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const T pv = p.v[c], qv = q.v[c];
        p.v[c] = pv + h.lr * (e * qv - h.regU * pv);
        q.v[c] = qv + h.lr * (e * pv - h.regI * qv);
        reg_part += (h.regU * pv) * pv + (h.regI * qv) * qv;
    }
    acc_reg += (double)reg_part;
    return l;
}
template <typename T, int MAXC, bool BCL>
__device__ __forceinline__ T camfc_step(RowVec<T, MAXC> &p, RowVec<T, MAXC> &q, T &bu, T &bj, T &bcreg, T *s_bc, const T rr,
                                        const unsigned long long pc, const unsigned long long pc2, const int dmax, const int lane,
                                        const PipeConst<T> &h, double &acc_reg, double &acc_ctx) {
    switch (dmax) {
    case 1: return camfc_step_dm<T, MAXC, 1, BCL>(p, q, bu, bj, bcreg, s_bc, rr, pc, pc2, lane, h, acc_reg, acc_ctx);
    case 2: return camfc_step_dm<T, MAXC, 2, BCL>(p, q, bu, bj, bcreg, s_bc, rr, pc, pc2, lane, h, acc_reg, acc_ctx);
    case 3: return camfc_step_dm<T, MAXC, 3, BCL>(p, q, bu, bj, bcreg, s_bc, rr, pc, pc2, lane, h, acc_reg, acc_ctx);
    case 4: return camfc_step_dm<T, MAXC, 4, BCL>(p, q, bu, bj, bcreg, s_bc, rr, pc, pc2, lane, h, acc_reg, acc_ctx);
    case 5: return camfc_step_dm<T, MAXC, 5, BCL>(p, q, bu, bj, bcreg, s_bc, rr, pc, pc2, lane, h, acc_reg, acc_ctx);
    case 6: return camfc_step_dm<T, MAXC, 6, BCL>(p, q, bu, bj, bcreg, s_bc, rr, pc, pc2, lane, h, acc_reg, acc_ctx);
    case 7: return camfc_step_dm<T, MAXC, 7, BCL>(p, q, bu, bj, bcreg, s_bc, rr, pc, pc2, lane, h, acc_reg, acc_ctx);
    default: return camfc_step_dm<T, MAXC, 8, BCL>(p, q, bu, bj, bcreg, s_bc, rr, pc, pc2, lane, h, acc_reg, acc_ctx);
    }
}

constexpr int CAMFC_PIPE_MAX_CONDS = 1024; // LDS form of condBias

template <typename T, int MAXC, bool FULL, int D, bool BCL>
__global__ __launch_bounds__(64) void sgd_camfc_pipe(SgdArgs<T> a, int64_t n, double *loss_out) {
    __shared__ RowVec<T, MAXC> h_p[D][64], h_q[D][64]; // the updated rows of the last D tuples (slot = t mod D)
    __shared__ T s_bc[BCL ? CAMFC_PIPE_MAX_CONDS : 1];
    const int lane = threadIdx.x;
    const int k = a.k, dmax = a.dmax;
    const HParams hp = *a.hp;
    const PipeConst<T> h = {(T)hp.lr, (T)hp.regU, (T)hp.regI, (T)hp.regB, (T)hp.regC, (T)hp.gm};
    T bcreg = (!BCL && lane < a.n_conds) ? a.condBias[lane] : (T)0;
    if (BCL)
        for (int c = lane; c < a.n_conds; c += 64) s_bc[c] = a.condBias[c];
    double loss = 0.0, acc_reg = 0.0, acc_ctx = 0.0;
    const int64_t n_main = n & ~(int64_t)63;

    // chunk staging: lane l holds tuple base + l
    // packed condition ids of tuple t: 8-bit fields of one word (register form), 16-bit fields of two words (LDS form)
    auto pack_ids = [&](int64_t t, unsigned long long &w, unsigned long long &w2) {
        w = 0;
        w2 = BCL ? ~0ull : 0ull;
        if (BCL) w = ~0ull;
        for (int d = 0; d < dmax; ++d) {
            const int c = a.sconds[t * dmax + d];
            if (BCL) {
                const unsigned long long f = (unsigned long long)(c < 0 ? 0xffff : (c & 0xffff)) << (16 * (d & 3));
                const unsigned long long m = ~(0xffffull << (16 * (d & 3)));
                if (d < 4) w = (w & m) | f;
                else w2 = (w2 & m) | f;
            } else {
                w |= (unsigned long long)(c < 0 ? 0xff : (c & 0xff)) << (8 * d);
            }
        }
    };
    auto load_ids = [&](int64_t base, int &mu, int &mj, T &mr, unsigned long long &mc, unsigned long long &mcb) {
        mu = a.su[base + lane];
        mj = a.sj[base + lane];
        mr = a.sr[base + lane];
        pack_ids(base + lane, mc, mcb);
    };

    if (n_main > 0) {
        int mu, mj, mu2, mj2;
        T mr, mr2;
        unsigned long long mc, mc2, mcb = 0, mcb2 = 0;
        load_ids(0, mu, mj, mr, mc, mcb);
        RowVec<T, MAXC> pn[D], qn[D];
        T bun[D], bjn[D];
#pragma unroll
        for (int s = 0; s < D; ++s) { // requests of the first D tuples
            const int u0 = prl(mu, s), j0 = prl(mj, s);
            pn[s] = load_row<T, MAXC, FULL>(a.P, u0, k, lane);
            qn[s] = load_row<T, MAXC, FULL>(a.Q, j0, k, lane);
            bun[s] = a.userBias[u0];
            bjn[s] = a.itemBias[j0];
        }
        RowVec<T, MAXC> p, q;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) p.v[c] = q.v[c] = (T)0;
        T bu = 0, bj = 0;
        int cu = -1, cj = -1;
        int hist_u = -1, hist_j = -1; // lane s: user / item of the tuple last processed in ring slot s
        T hist_bu = 0, hist_bj = 0;   // lane s: its updated scalar biases
        for (int64_t base = 0; base < n_main; base += 64) {
            // the next chunk's ids, requested a chunk ahead and unconditionally (the last chunk re-requests itself: its look-ahead past
            // the end then re-requests rows of this chunk, which are never used)
            load_ids(base + 64 < n_main ? base + 64 : base, mu2, mj2, mr2, mc2, mcb2);
            T vl = 0; // lane i: the uniform loss terms of the chunk's tuple i
            auto stage = [&](const int i, const int s, const int nu, const int nj) __attribute__((always_inline)) {
                const int uu = prl(mu, i), jj = prl(mj, i);
                const T rr = prl(mr, i);
                const unsigned long long pc = prl(mc, i), pcb = BCL ? prl(mcb, i) : 0ull;
                // ---- the rows requested D tuples ago, then the request for tuple t + D into the same slot
                const RowVec<T, MAXC> cand_p = pn[s], cand_q = qn[s];
                const T cand_bu = bun[s], cand_bj = bjn[s];
                pn[s] = load_row<T, MAXC, FULL>(a.P, nu, k, lane);
                qn[s] = load_row<T, MAXC, FULL>(a.Q, nj, k, lane);
                bun[s] = a.userBias[nu];
                bjn[s] = a.itemBias[nj];
                // ---- which copy is current: registers (same row as the previous tuple), the LDS ring (written by one of the last D
                // tuples), or the requested one
                if (uu != cu) {
                    const unsigned long long m = __ballot(hist_u == uu) & ((1ull << D) - 1);
                    if (m) { // newest writer: ages 1..D <-> slots s-1, s-2, ..., s-D (mod D)
                        int slot = 0;
#pragma unroll
                        for (int age = D; age >= 1; --age) {
                            const int sl = ((s - age) % D + D) % D;
                            if ((m >> sl) & 1ull) slot = sl;
                        }
                        p = h_p[slot][lane];
                        bu = prl(hist_bu, slot);
                    } else {
                        p = cand_p;
                        bu = cand_bu;
                    }
                }
                if (jj != cj) {
                    const unsigned long long m = __ballot(hist_j == jj) & ((1ull << D) - 1);
                    if (m) {
                        int slot = 0;
#pragma unroll
                        for (int age = D; age >= 1; --age) {
                            const int sl = ((s - age) % D + D) % D;
                            if ((m >> sl) & 1ull) slot = sl;
                        }
                        q = h_q[slot][lane];
                        bj = prl(hist_bj, slot);
                    } else {
                        q = cand_q;
                        bj = cand_bj;
                    }
                }
                cu = uu;
                cj = jj;
                const T l_t = camfc_step<T, MAXC, BCL>(p, q, bu, bj, bcreg, s_bc, rr, pc, pcb, dmax, lane, h, acc_reg, acc_ctx);
                vl = lane == i ? l_t : vl;
                // ---- publish: HBM, the LDS ring, the id / bias history
                store_row<T, MAXC, FULL>(a.P, uu, k, lane, p);
                store_row<T, MAXC, FULL>(a.Q, jj, k, lane, q);
                a.userBias[uu] = bu; // every lane stores the same value to the same word: no branch in the loop
                a.itemBias[jj] = bj;
                h_p[s][lane] = p;
                h_q[s][lane] = q;
                hist_u = pwl(hist_u, uu, s, lane);
                hist_j = pwl(hist_j, jj, s, lane);
                hist_bu = pwl(hist_bu, bu, s, lane);
                hist_bj = pwl(hist_bj, bj, s, lane);

            };
            // rounds of D tuples; all but the last round of a chunk look ahead inside the chunk, the last one into the next chunk's ids
            // (peeled, so that the next chunk's id request is first touched 64 - D tuples after it was issued)
            for (int i0 = 0; i0 < 64 - D; i0 += D) {
#pragma unroll
                for (int s = 0; s < D; ++s) stage(i0 + s, s, prl(mu, i0 + s + D), prl(mj, i0 + s + D));
            }
#pragma unroll
            for (int s = 0; s < D; ++s) stage(64 - D + s, s, prl(mu2, s), prl(mj2, s));
            loss += pwave_sum((double)vl);
            mu = mu2;
            mj = mj2;
            mr = mr2;
            mc = mc2;
            mcb = mcb2;
        }
    }
    // ---- the last n mod 64 tuples: request after the previous tuple's stores, one by one
    for (int64_t t = n_main; t < n; ++t) {
        const int uu = a.su[t], jj = a.sj[t];
        const T rr = a.sr[t];
        unsigned long long pc, pcb;
        pack_ids(t, pc, pcb);
        RowVec<T, MAXC> p = load_row<T, MAXC, FULL>(a.P, uu, k, lane), q = load_row<T, MAXC, FULL>(a.Q, jj, k, lane);
        T bu = a.userBias[uu], bj = a.itemBias[jj];
        loss += (double)camfc_step<T, MAXC, BCL>(p, q, bu, bj, bcreg, s_bc, rr, pc, pcb, dmax, lane, h, acc_reg, acc_ctx);
        store_row<T, MAXC, FULL>(a.P, uu, k, lane, p);
        store_row<T, MAXC, FULL>(a.Q, jj, k, lane, q);
        a.userBias[uu] = bu;
        a.itemBias[jj] = bj;
    }
    if (BCL) {
        for (int c = lane; c < a.n_conds; c += 64) a.condBias[c] = s_bc[c];
    } else if (lane < a.n_conds) {
        a.condBias[lane] = bcreg;
    }
    loss += pwave_sum(acc_reg) + (double)h.regB * pwave_sum(acc_ctx);
    if (lane == 0) loss_out[0] = loss * 0.5;
}

} // namespace

// <= 8 context dimensions, k <= 256; <= 64 conditions: condBias in a register (one per lane, one byte per id); <= 1024: condBias in LDS
// (round 4: the real Frappe file has 343)
bool camfc_pipe_supported(int k, int n_conds, int dmax) {
    return k >= 1 && k <= 256 && n_conds <= CAMFC_PIPE_MAX_CONDS && dmax >= 1 && dmax <= 8 && !getenv("CMI_NO_CAMFC_PIPE");
}

template <typename T, bool BCL>
static hipError_t launch_camfc_pipe_bc(const SgdArgs<T> &a, int64_t n, double *loss_out, hipStream_t s) {
    const int k = a.k;
    // ring depth: the requests of tuple t are the oldest of 4 * rowops + ... outstanding operations when they are consumed; the
    // hardware counts 64 of them, so D * (2 * rowops + 2) * 2 stays below that
    // (D divides the 64-tuple chunk)
    if (k == 64) hipLaunchKernelGGL((sgd_camfc_pipe<T, 1, true, 8, BCL>), dim3(1), dim3(64), 0, s, a, n, loss_out);
    else if (k == 128) hipLaunchKernelGGL((sgd_camfc_pipe<T, 2, true, 8, BCL>), dim3(1), dim3(64), 0, s, a, n, loss_out);
    else if (k == 256) hipLaunchKernelGGL((sgd_camfc_pipe<T, 4, true, sizeof(T) == 8 ? 4 : 8, BCL>), dim3(1), dim3(64), 0, s, a, n, loss_out);
    else if (k < 64) hipLaunchKernelGGL((sgd_camfc_pipe<T, 1, false, 8, BCL>), dim3(1), dim3(64), 0, s, a, n, loss_out);
    else if (k < 128) hipLaunchKernelGGL((sgd_camfc_pipe<T, 2, false, 4, BCL>), dim3(1), dim3(64), 0, s, a, n, loss_out);
    else hipLaunchKernelGGL((sgd_camfc_pipe<T, 4, false, 2, BCL>), dim3(1), dim3(64), 0, s, a, n, loss_out);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_camfc_pipe(const SgdArgs<T> &a, int64_t n, double *loss_out, hipStream_t s) {
    return a.n_conds <= 64 ? launch_camfc_pipe_bc<T, false>(a, n, loss_out, s) : launch_camfc_pipe_bc<T, true>(a, n, loss_out, s);
}
template hipError_t launch_camfc_pipe<float>(const SgdArgs<float> &, int64_t, double *, hipStream_t);
template hipError_t launch_camfc_pipe<double>(const SgdArgs<double> &, int64_t, double *, hipStream_t);

} // namespace cmi

// svdpp_team.hip -- SVD++ (src/carskit/alg/baseline/cf/SVDPlusPlus.java:58-128) with a workgroup per dependency chain link.
//
// Every rating (u, j) of SVD++ reads the implicit-feedback rows Y[i] of ALL items i in N(u) (the items u rated), and writes every one of
// them: no two ratings commute, the epoch is one chain in the 2-D train matrix's row-major order, and a link is O(|N(u)| k) work.
// ext_serial_wave (ext_kernels.hip) walks that work with one wave and a dependent global access per row: 17.6 / 16.5 / 25.7 us per rating at
// k = 10 / 64 / 128 on a 59 K-rating train matrix with |N(u)| = 30 (one CPU core: 0.31 / 2.1 / 4.7 us); this kernel: 1.5 / 1.9 / 2.6 us
// (tests/tools/bench_ext_models.py).  What the order leaves to exploit:
//   * the ratings of one user are consecutive (row-major order) and all of them touch the SAME rows Y[N(u)] and P[u]: the rows are loaded
//     into LDS once per user, updated there by every rating of the user, and written back once;
//   * inside a link everything is wide: |N(u)| dot products <Y[i], Q[j]> (one 16-lane group per row, DPP row sums), the per-factor column
//     sums over Y[N(u)] (thread f adds the rows in item order -- the reference's own summation order), and |N(u)| k element updates spread
//     over the 1024 threads.  Three workgroup barriers per link; the error e is recomputed by every wave from the same LDS partials (the
//     same tree in every wave), so it needs no broadcast;
//   * Q[j] of the NEXT rating is requested while this one is computed (the items of one user are distinct, so it cannot be stale).
// Per element the expressions are ext_serial_wave's; dots and the prediction are tree sums.  A user whose rows do not fit the LDS budget
// (or whose ratings do not arrive as one run) is walked by wave 0 with ext_serial_wave's code.
#include "mf_sgd_kernels.hpp"
#include "env_knobs.hpp"
#include "sgd_device.hpp"

#include <cmath>
#include <cstdlib>

namespace cmi {
namespace {

template <typename T>
__device__ __forceinline__ T wsum(T x) { // wave sum, every lane gets the same total (fixed xor tree; the fallback path and the epilogue)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}
// wave sum on the link's critical path: DPP row rotations + row broadcasts (60 cycles instead of six ds_bpermute round trips), result
// taken from lane 63 as a uniform value
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float tdpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double tdpp(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float trl(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
__device__ __forceinline__ double trl(double x, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
template <typename T>
__device__ __forceinline__ T wsum_dpp(T x) {
    x += tdpp<0x128, 0xf>(x);
    x += tdpp<0x124, 0xf>(x);
    x += tdpp<0x122, 0xf>(x);
    x += tdpp<0x121, 0xf>(x);
    x += tdpp<0x142, 0xa>(x);
    x += tdpp<0x143, 0xc>(x);
    return trl(x, 63);
}
// a barrier that orders LDS traffic only: the requests for the next rating's row stay in flight across it (__syncthreads would drain
// them); inside a user's run no wave reads HBM data another wave wrote
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float gsum16(float x) { return row_sum16(x); }
__device__ __forceinline__ double gsum16(double x) { return row_sum16(x); }

// one rating, one wave, rows in HBM: the fallback (same code as ext_serial_wave's SVD++ branch)
template <typename T>
__device__ void svdpp_link_wave(const ExtArgs<T> &a, int64_t t, int lane, const T lr, const T regU, const T regI, const T regB, const T gm,
                                double &loss, double &lpart) {
    const int k = a.k;
    const int uu = a.su[t], jj = a.sj[t];
    const T rr = a.sr[t];
    T *pu = a.P + (size_t)uu * k, *qj = a.Q + (size_t)jj * k;
    const int32_t b = a.ui_ptr[uu], en = a.ui_ptr[uu + 1];
    const T w = (T)sqrt((double)(en - b));
    const T bu = a.userBias[uu], bj = a.itemBias[jj];
    T part = 0;
    for (int f = lane; f < k; f += 64) part += pu[f] * qj[f];
    T pred = gm + bu + bj + wsum(part);
    for (int32_t q = b; q < en; ++q) {
        const T *y = a.Y + (size_t)a.ui_items[q] * k;
        T pp = 0;
        for (int f = lane; f < k; f += 64) pp += y[f] * qj[f];
        pred += wsum(pp) / w;
    }
    const T e = rr - pred;
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS: s_Y[max_rows][k] | s_p[k] | s_q[k] | s_part[66] | s_rr[max_rows] (T) | s_items[max_rows] | s_j[max_rows] (int32)
template <typename T, int NT> // NT threads: 16-lane groups NT/16, waves NT/64
__global__ __launch_bounds__(NT) void svdpp_team(ExtArgs<T> a, int64_t n, int max_rows, double *loss_out) {
    extern __shared__ unsigned char smem_raw[];
    const int k = a.k;
    T *s_Y = reinterpret_cast<T *>(smem_raw);
    T *s_p = s_Y + (size_t)max_rows * k;
    T *s_q = s_p + k;
    T *s_part = s_q + k;
    T *s_rr = s_part + 66;
    int32_t *s_items = reinterpret_cast<int32_t *>(s_rr + max_rows);
    int32_t *s_j = s_items + max_rows;
    __shared__ double s_loss[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = tid & 15, g = tid >> 4;
    const HParams hp = *a.hp;
    const T lr = (T)hp.lr, regU = (T)hp.regU, regI = (T)hp.regI, regB = (T)hp.regB, gm = (T)hp.gm;
    constexpr int NG = NT / 16;
    const int r_step = NT / k, f_step = NT - r_step * k; // (row, factor) of element idx + NT
    if (tid < 64) s_part[tid] = (T)0; // slots of groups that do not exist stay zero
    const int r0 = tid / k, f0 = tid - r0 * k;                // (row, factor) of element tid
    double loss = 0.0;  // uniform terms: counted by thread 0
    double lpart = 0.0; // element terms of this thread
    int64_t t = 0;
    while (t < n) {
        const int uu = a.su[t];
        const int32_t b = a.ui_ptr[uu], en = a.ui_ptr[uu + 1];
        const int cnt = en - b;
        // the run of this user's ratings: in the 2-D train matrix it is exactly the user's row, |N(u)| ratings (checked; else scanned)
        int64_t t_end = t + cnt;
        if (cnt <= 0 || t_end > n || a.su[t_end - 1] != uu || (t_end < n && a.su[t_end] == uu)) {
            t_end = t + 1;
            while (t_end < n && a.su[t_end] == uu) ++t_end;
        }
        const int run = (int)(t_end - t);
        // a row listed twice in N(u) (a (user, item) pair given twice: not a train matrix, but the C ABI takes any tuple list) is updated
        // twice per rating by the sequential loop, the second time from the first result: two LDS copies cannot do that
        bool twice = false;
        if (cnt > 0 && cnt <= max_rows) {
            for (int x = tid; x < cnt; x += NT) s_items[x] = a.ui_items[b + x];
            __syncthreads();
            bool mine = false;
            for (int x = tid; x + 1 < cnt; x += NT) mine = mine || s_items[x] == s_items[x + 1];
            twice = __syncthreads_or(mine);
        }
        if (cnt > max_rows || cnt <= 0 || run > max_rows || twice) { // rows do not fit (or repeat): wave 0 walks the run through HBM
            if (wave == 0) {
                double l0 = 0.0;
                for (int64_t x = t; x < t_end; ++x) svdpp_link_wave<T>(a, x, lane, lr, regU, regI, regB, gm, l0, lpart);
                if (lane == 0) loss += l0;
            }
            __syncthreads();
            t = t_end;
            continue;
        }
        // ---- load the user's rows and the run's (item, rating) pairs: every request is issued before the first wait
        {
            int r = r0, f = f0;
            for (int x = tid; x < cnt * k; x += NT) {
                s_Y[x] = a.Y[(size_t)a.ui_items[b + r] * k + f];
                r += r_step;
                f += f_step;
                if (f >= k) {
                    f -= k;
                    ++r;
                }
            }
        }
        for (int x = tid; x < run; x += NT) {
            s_j[x] = a.sj[t + x];
            s_rr[x] = a.sr[t + x];
        }
        if (tid < k) s_p[tid] = a.P[(size_t)uu * k + tid];
        const T w = (T)sqrt((double)cnt);
        T bu = a.userBias[uu];
        // the rating about to run: Q[j][tid] and itemBias[j] -- requested one rating ahead
        const int j_first = a.sj[t];
        T q_next = tid < k ? a.Q[(size_t)j_first * k + tid] : (T)0;
        T bj_next = a.itemBias[j_first];
        __syncthreads();
        for (int x = 0; x < run; ++x) {
            const int jj = s_j[x];
            const T rr = s_rr[x], bj = bj_next;
            // ---- P1: Q[j] into LDS, the next rating's row requested
            if (tid < k) s_q[tid] = q_next;
            // (a stream that repeats one (user, item) pair back to back -- not a train matrix, but the C ABI takes any tuple list -- would
            // make this request stale: it is then issued after this rating's stores instead)
            const int jn = s_j[x + 1 < run ? x + 1 : x];
            const bool repeat = jn == jj && x + 1 < run;
            if (!repeat) {
                if (tid < k) q_next = a.Q[(size_t)jn * k + tid];
                bj_next = a.itemBias[jn];
            }
            lds_barrier();
            // ---- P2: one 16-lane group per row: <Y[i], Q[j]> / w; group 0 adds <P[u], Q[j]>; thread f: the column sum in item order
            T acc = 0;
            for (int r = g; r < cnt; r += NG) {
                T part = 0;
                for (int f = l16; f < k; f += 16) part += s_Y[(size_t)r * k + f] * s_q[f];
                acc += gsum16(part) / w;
            }
            if (l16 == 0) s_part[g] = acc;
            if (g == 0) {
                T part = 0;
                for (int f = l16; f < k; f += 16) part += s_p[f] * s_q[f];
                const T d = gsum16(part);
                if (l16 == 0) s_part[64] = d;
            }
            T sum_f = 0;
            if (tid < k) { // eight rows requested at a time, added one by one in item order (the reference's sum_f loop)
                int r = 0;
                for (; r + 8 <= cnt; r += 8) {
                    T v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = s_Y[(size_t)(r + i) * k + tid];
#pragma unroll
                    for (int i = 0; i < 8; ++i) sum_f += v[i];
                }
                for (; r < cnt; ++r) sum_f += s_Y[(size_t)r * k + tid];
            }
            lds_barrier();
            // ---- P3: every wave forms the same e from the same partials; then all element updates
            const T ytot = wsum_dpp(s_part[lane]);
            const T pred = gm + bu + bj + s_part[64] + ytot;
            const T e = rr - pred;
            if (tid == 0) {
                loss += (double)(e * e) + (double)((regB * bu) * bu) + (double)((regB * bj) * bj);
                a.itemBias[jj] = bj + lr * (e - regB * bj);
            }
            bu = bu + lr * (e - regB * bu); // every thread keeps the user's bias; thread 0 stores it at the end of the run
            if (tid < k) {
                const T sum_ys = w > (T)0 ? sum_f / w : sum_f;
                const T puf = s_p[tid], qjf = s_q[tid];
                s_p[tid] = puf + lr * (e * qjf - regU * puf);
                a.Q[(size_t)jj * k + tid] = qjf + lr * (e * (puf + sum_ys) - regI * qjf);
                lpart += (double)((regU * puf) * puf + (regI * qjf) * qjf);
            }
            {
                int f = f0;
                for (int idx = tid; idx < cnt * k; idx += NT) {
                    const T ykf = s_Y[idx];
                    s_Y[idx] = ykf + lr * ((e * s_q[f]) / w - regU * ykf);
                    lpart += (double)((regU * ykf) * ykf);
                    f += f_step;
                    if (f >= k) f -= k;
                }
            }
            if (repeat) {
                __syncthreads(); // drains this rating's stores: the row below is the updated one
                if (tid < k) q_next = a.Q[(size_t)jn * k + tid];
                bj_next = a.itemBias[jn];
            }
            lds_barrier(); // s_q and s_part are rewritten by the next rating; s_Y / s_p updates are visible
        }
        // ---- write the user's rows back
        {
            int r = r0, f = f0;
            for (int x = tid; x < cnt * k; x += NT) {
                a.Y[(size_t)s_items[r] * k + f] = s_Y[x];
                r += r_step;
                f += f_step;
                if (f >= k) {
                    f -= k;
                    ++r;
                }
            }
        }
        if (tid < k) a.P[(size_t)uu * k + tid] = s_p[tid];
        if (tid == 0) a.userBias[uu] = bu;
        __syncthreads(); // the next user's gather sees these rows (workgroup scope), and the LDS arrays are free
        t = t_end;
    }
    const double wl = wsum(lpart);
    if (lane == 0) s_loss[wave] = wl;
    __syncthreads();
    if (tid == 0) {
        double total = loss;
        for (int i = 0; i < NT / 64; ++i) total += s_loss[i];
        loss_out[0] = total * 0.5;
    }
}

} // namespace

// unused s_part lanes must read as zero: the kernel sums all 64 slots
template <typename T>
hipError_t launch_svdpp_team(const ExtArgs<T> &a, int64_t n, double *loss_out, hipStream_t s) {
    const size_t budget = 144 * 1024; // of the CU's 160 KB
    const size_t fixed = ((size_t)2 * a.k + 66) * sizeof(T);
    const size_t per_row = (size_t)a.k * sizeof(T) + sizeof(T) + 2 * sizeof(int32_t);
    size_t rows = (budget - fixed) / per_row;
    if (rows > 4096) rows = 4096;
    const size_t lds = fixed + rows * per_row + 16;
    // team size: 1024 threads measured fastest (measured on a stream of one-rating runs: 2.9 / 3.8 / 5.1 us per rating at k = 10 / 64 / 128 against 3.2 / 5.7 / 9.1 with
    // 256 threads) although a link is only ~375 instructions per wave: the wide parts (|N(u)| k element updates, one group per row) win more
    // from 16 waves than the uniform part loses; CMI_SVDPP_THREADS=256|512 for A/B runs
    const char *env = cmi_exp_env("CMI_SVDPP_THREADS");
    int nt = env ? atoi(env) : 1024;
    if (nt < a.k) nt = a.k <= 256 ? 256 : a.k <= 512 ? 512 : 1024;
#define CMI_SVDPP_LAUNCH(NTV)                                                                                                             \
    do {                                                                                                                                  \
        auto fn = svdpp_team<T, NTV>;                                                                                                     \
        if (lds > 64 * 1024)                                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
        hipLaunchKernelGGL(fn, dim3(1), dim3(NTV), lds, s, a, n, (int)rows, loss_out);                                                    \
    } while (0)
    if (nt <= 256) CMI_SVDPP_LAUNCH(256);
    else if (nt <= 512) CMI_SVDPP_LAUNCH(512);
    else CMI_SVDPP_LAUNCH(1024);
#undef CMI_SVDPP_LAUNCH
    return hipGetLastError();
}
template hipError_t launch_svdpp_team<float>(const ExtArgs<float> &, int64_t, double *, hipStream_t);
template hipError_t launch_svdpp_team<double>(const ExtArgs<double> &, int64_t, double *, hipStream_t);

bool svdpp_team_supported(int k) { return k >= 1 && k <= 1024 && !getenv("CMI_NO_SVDPP_TEAM"); }

} // namespace cmi

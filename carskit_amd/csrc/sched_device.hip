// sched_device.hip -- the hub-chain level schedule (level_schedule.cpp build_chain_schedule) and the spoke arena's position lists
// (cmi_api.cpp arena_positions) built ON THE DEVICE.  Same results, element for element, as the host builders they replace: the
// schedule is the order-exact restatement of librec's MatrixIterator order (CAMF_CI.java:80 `for (MatrixEntry me :
// trainMatrix)`), so nothing about it may change -- only where it is computed.  With epochs at milliseconds the host walk (11 ns per
// tuple, one core: 0.4 s for C3's 50 M tuples, 2 s for north_star's 200 M) was the long step of a run.
//
// The recurrence (chain_pass): tuple t, in CRS order, with hub row h and spoke row s:
//     A = level of h's previous tuple, B = level of s's previous tuple
//     A > B and h's current unit has < max_chain tuples  ->  t joins that unit (level A)
//     else                                               ->  t starts a unit at level max(A, B) + 1
// is sequential in t only through those two predecessors.  On the device ONE LANE OWNS A HUB ROW for the whole walk (its level, unit
// and unit length stay in registers) and walks the row's tuples in CRS order; the spoke side is a table of 64-bit words
// {tuples of the row done so far, level of the last one}, read and written with single atomic accesses: a lane may take tuple t once
// the spoke's word counts exactly the tuples that precede t in the spoke's own CRS chain (`want`, the rank of t among them).  The tuple
// with the smallest CRS index not yet taken always finds both predecessors done, so the walk cannot deadlock as long as every hub row
// has a lane that keeps trying: a lane owns up to HMAX rows and tries them in turn (never spinning on one: the lanes of a wave run in
// lockstep), and the grid is sized to be resident as a whole.  As a backstop a wave that sees no progress for a long time parks its
// state and exits, and the host launches again.
//
// Everything around the walk is sorting and scanning (rocPRIM): tuples by hub row (the rows' CRS chains) and by spoke row (the ranks),
// units by (level, length descending, first tuple) -- the host's counting sort order -- and prefix sums for the offsets.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "level_schedule.hpp"
#include "sched_device.hpp"

namespace cmi {

namespace {

struct DevPool { // device allocations of one build, freed together
    std::vector<void *> ptrs;
    hipError_t err = hipSuccess;
    template <typename T>
    T *get(size_t count) {
        if (err != hipSuccess) return nullptr;
        void *p = nullptr;
        err = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if (err != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return (T *)p;
    }
    ~DevPool() {
        for (void *p : ptrs) (void)hipFree(p);
    }
};

__global__ void k_iota(int32_t *v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = (int32_t)i;
}
__global__ void k_hist(const int32_t *key, int64_t n, int32_t *cnt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) atomicAdd(&cnt[key[i]], 1);
}
// sorted by key (stable): rank[t] = position of t inside its key's segment
__global__ void k_rank(const int32_t *skey, const int32_t *st, const int32_t *off, int64_t n, int32_t *rank) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        rank[st[i]] = (int32_t)(i - off[skey[i]]);
}
// the hub rows' lists: spoke row and rank of every list entry
__global__ void k_lists(const int32_t *lt, const int32_t *spoke_of, const int32_t *rank, int64_t n, int32_t *sp_h, int32_t *want_h) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t t = lt[i];
        sp_h[i] = spoke_of[t];
        want_h[i] = rank[t];
    }
}
__global__ void k_init_state(const int32_t *hub_off, int32_t n_hub, int32_t *sv) { // sv[4 i ..] = pos, level, unit, len
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_hub; i += (int64_t)gridDim.x * blockDim.x) {
        sv[4 * i] = hub_off[i];
        sv[4 * i + 1] = 0;
        sv[4 * i + 2] = -1;
        sv[4 * i + 3] = 0;
    }
}

struct WalkArgs {
    const int32_t *lt, *sp_h, *want_h, *hub_off; // the hub rows' lists (tuple, spoke row, rank) and their offsets
    int32_t n_hub, max_chain;
    unsigned long long *spoke; // per spoke row: tuples done << 32 | level of the last one
    int32_t *sv;               // parked lane state per hub row (k_init_state)
    // outputs (null: count only)
    int32_t *unit_of;    // per tuple: CRS index of its unit's first tuple
    uint8_t *pos;        // per tuple: position inside its unit
    int32_t *unit_level; // per tuple: level of the unit it STARTS (0: it starts none)
    uint8_t *unit_len;   // per tuple: length of the unit it starts
    unsigned long long *counters; // [0] units, [1] max level, [2] tuples left after this launch
    int idle_limit;
};

template <int HMAX>
__global__ __launch_bounds__(256) void k_walk(WalkArgs a) {
    const int64_t T = (int64_t)gridDim.x * blockDim.x, g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int32_t pos[HMAX], end[HMAX], lev[HMAX], unit[HMAX], len[HMAX];
    unsigned long long units = 0;
    int32_t top = 0;
#pragma unroll
    for (int k = 0; k < HMAX; ++k) {
        const int64_t h = g + (int64_t)k * T;
        if (h < a.n_hub) {
            pos[k] = a.sv[4 * h];
            lev[k] = a.sv[4 * h + 1];
            unit[k] = a.sv[4 * h + 2];
            len[k] = a.sv[4 * h + 3];
            end[k] = a.hub_off[h + 1];
        } else {
            pos[k] = end[k] = 0;
            lev[k] = len[k] = 0;
            unit[k] = -1;
        }
    }
    int idle = 0;
    for (;;) {
        bool left = false, progressed = false;
#pragma unroll
        for (int k = 0; k < HMAX; ++k) {
            if (pos[k] >= end[k]) continue;
            left = true;
            const int32_t sp = a.sp_h[pos[k]], want = a.want_h[pos[k]];
            const unsigned long long v = __hip_atomic_load(&a.spoke[sp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int32_t)(v >> 32) != want) continue; // the spoke row's previous tuple has not been taken yet: try the next row
            const int32_t t = a.lt[pos[k]], A = lev[k], B = (int32_t)(v & 0xFFFFFFFFu);
            int32_t l, p;
            if (A > B && len[k] < a.max_chain) {
                l = A;
                p = len[k]++;
                if (a.unit_len) a.unit_len[unit[k]] = (uint8_t)len[k];
            } else {
                l = (A > B ? A : B) + 1;
                p = 0;
                len[k] = 1;
                unit[k] = t;
                ++units;
                if (a.unit_level) {
                    a.unit_level[t] = l;
                    a.unit_len[t] = 1;
                }
            }
            lev[k] = l;
            top = l > top ? l : top;
            __hip_atomic_store(&a.spoke[sp], ((unsigned long long)(uint32_t)(want + 1) << 32) | (uint32_t)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.unit_of) {
                a.unit_of[t] = unit[k];
                a.pos[t] = (uint8_t)p;
            }
            ++pos[k];
            progressed = true;
        }
        if (__builtin_amdgcn_ballot_w64(left) == 0) break;
        if (__builtin_amdgcn_ballot_w64(progressed) == 0) {
            if (++idle > a.idle_limit) break; // park and let the host launch again (other kernels on the device, a grid that is not resident)
            __builtin_amdgcn_s_sleep(8);
        } else idle = 0;
    }
    unsigned long long remaining = 0;
#pragma unroll
    for (int k = 0; k < HMAX; ++k) {
        const int64_t h = g + (int64_t)k * T;
        if (h < a.n_hub) {
            a.sv[4 * h] = pos[k];
            a.sv[4 * h + 1] = lev[k];
            a.sv[4 * h + 2] = unit[k];
            a.sv[4 * h + 3] = len[k];
            remaining += (unsigned long long)(end[k] - pos[k]);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        units += __shfl_xor(units, m, 64);
        remaining += __shfl_xor(remaining, m, 64);
        const int32_t o = __shfl_xor(top, m, 64);
        top = o > top ? o : top;
    }
    if ((threadIdx.x & 63) == 0) {
        if (units) atomicAdd(&a.counters[0], units);
        if (top) atomicMax(&a.counters[1], (unsigned long long)top);
        if (remaining) atomicAdd(&a.counters[2], remaining);
    }
}

// The same walk for sides with more hub rows than 16 per resident lane (10 M users): a lane sweeps rows g, g + T, g + 2T, ... with the
// rows' state in memory (sv) instead of registers -- one try per row and sweep.
__global__ __launch_bounds__(256) void k_walk_mem(WalkArgs a) {
    const int64_t T = (int64_t)gridDim.x * blockDim.x, g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long units = 0, remaining = 0;
    int32_t top = 0;
    int idle = 0;
    for (;;) {
        bool left = false, progressed = false;
        for (int64_t h = g; h < a.n_hub; h += T) {
            int32_t pos = a.sv[4 * h];
            const int32_t end = a.hub_off[h + 1];
            if (pos >= end) continue;
            int32_t lev = a.sv[4 * h + 1], unit = a.sv[4 * h + 2], len = a.sv[4 * h + 3];
            bool moved = false;
            while (pos < end) { // as far as the row gets in this sweep
                const int32_t sp = a.sp_h[pos], want = a.want_h[pos];
                const unsigned long long v = __hip_atomic_load(&a.spoke[sp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int32_t)(v >> 32) != want) break;
                const int32_t t = a.lt[pos], B = (int32_t)(v & 0xFFFFFFFFu);
                int32_t l, p;
                if (lev > B && len < a.max_chain) {
                    l = lev;
                    p = len++;
                    if (a.unit_len) a.unit_len[unit] = (uint8_t)len;
                } else {
                    l = (lev > B ? lev : B) + 1;
                    p = 0;
                    len = 1;
                    unit = t;
                    ++units;
                    if (a.unit_level) {
                        a.unit_level[t] = l;
                        a.unit_len[t] = 1;
                    }
                }
                lev = l;
                top = l > top ? l : top;
                __hip_atomic_store(&a.spoke[sp], ((unsigned long long)(uint32_t)(want + 1) << 32) | (uint32_t)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.unit_of) {
                    a.unit_of[t] = unit;
                    a.pos[t] = (uint8_t)p;
                }
                ++pos;
                moved = true;
            }
            if (moved) {
                a.sv[4 * h] = pos;
                a.sv[4 * h + 1] = lev;
                a.sv[4 * h + 2] = unit;
                a.sv[4 * h + 3] = len;
                progressed = true;
            }
            left = left || pos < end;
        }
        if (__builtin_amdgcn_ballot_w64(left) == 0) break;
        if (__builtin_amdgcn_ballot_w64(progressed) == 0) {
            if (++idle > a.idle_limit) break;
            __builtin_amdgcn_s_sleep(8);
        } else idle = 0;
    }
    for (int64_t h = g; h < a.n_hub; h += T) remaining += (unsigned long long)(a.hub_off[h + 1] - a.sv[4 * h]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        units += __shfl_xor(units, m, 64);
        remaining += __shfl_xor(remaining, m, 64);
        const int32_t o = __shfl_xor(top, m, 64);
        top = o > top ? o : top;
    }
    if ((threadIdx.x & 63) == 0) {
        if (units) atomicAdd(&a.counters[0], units);
        if (top) atomicMax(&a.counters[1], (unsigned long long)top);
        if (remaining) atomicAdd(&a.counters[2], remaining);
    }
}

__global__ void k_flags(const int32_t *unit_level, int64_t n, int32_t *flag) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) flag[i] = unit_level[i] != 0;
}
// dense unit ids in CRS order of the units' first tuples (the host walk's n_units++) and the sort key (level asc, length desc)
__global__ void k_units(const int32_t *unit_level, const uint8_t *unit_len, const int32_t *dense, int64_t n, int32_t max_chain, uint32_t *key,
                        int32_t *uid, uint8_t *ulen) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        if (unit_level[t] != 0) {
            const int32_t q = dense[t];
            key[q] = (uint32_t)(unit_level[t] - 1) * (uint32_t)max_chain + (uint32_t)(max_chain - unit_len[t]);
            uid[q] = q;
            ulen[q] = unit_len[t];
        }
}
__global__ void k_rank_units(const int32_t *sorted_uid, const uint8_t *ulen, int64_t nu, int32_t *rank_of, int32_t *len_sorted) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nu; r += (int64_t)gridDim.x * blockDim.x) {
        const int32_t q = sorted_uid[r];
        rank_of[q] = (int32_t)r;
        len_sorted[r] = ulen[q];
    }
}
__global__ void k_perm(const int32_t *unit_of, const uint8_t *pos, const int32_t *dense, const int32_t *rank_of, const int32_t *unit_off, int64_t n,
                       int32_t *perm) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        perm[unit_off[rank_of[dense[unit_of[t]]]] + pos[t]] = (int32_t)t;
}
// level_off[l] = units with level <= l = first sorted key >= l * max_chain
__global__ void k_level_off(const uint32_t *skey, int64_t nu, int32_t nl, int32_t max_chain, int64_t *level_off) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l > nl) return;
    const uint64_t want = (uint64_t)l * (uint64_t)max_chain;
    int64_t lo = 0, hi = nu;
    while (lo < hi) {
        const int64_t mid = (lo + hi) / 2;
        if ((uint64_t)skey[mid] < want) lo = mid + 1;
        else hi = mid;
    }
    level_off[l] = lo;
}

// next / first of the spoke arena out of the stream sorted by spoke row (stable: ascending positions inside a row)
__global__ void k_arena(const int32_t *skey, const int32_t *spos, int64_t n, int32_t *next, int32_t *first) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t r = skey[i];
        const bool head = i == 0 || skey[i - 1] != r, tail = i + 1 == n || skey[i + 1] != r;
        if (head) first[r] = spos[i];
        if (!tail) next[spos[i]] = spos[i + 1];
    }
}
__global__ void k_arena_wrap(const int32_t *skey, const int32_t *spos, int64_t n, int32_t *next, const int32_t *first) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (i + 1 == n || skey[i + 1] != skey[i]) next[spos[i]] = first[skey[i]]; // the row's last tuple wraps to its first
}
__global__ void k_fill(int32_t *v, int64_t n, int32_t x) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = x;
}

inline unsigned grid_for(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 8192)); }
inline unsigned bits_for(int64_t values) { // bits needed for keys in [0, values)
    unsigned b = 1;
    while (b < 32 && ((int64_t)1 << b) < values) ++b;
    return b;
}

#define SD(expr)                      \
    do {                              \
        const hipError_t e_ = (expr); \
        if (e_ != hipSuccess) return e_; \
    } while (0)

// stable sort of (key[t], t) for t < m by key; off[k] = first sorted position of key k
hipError_t sort_by(DevPool &pool, hipStream_t s, const int32_t *key, int64_t m, int32_t n_keys, int32_t *&skey, int32_t *&st, int32_t *&off) {
    int32_t *iota = pool.get<int32_t>((size_t)m), *cnt = pool.get<int32_t>((size_t)n_keys + 1);
    skey = pool.get<int32_t>((size_t)m);
    st = pool.get<int32_t>((size_t)m);
    off = pool.get<int32_t>((size_t)n_keys + 1);
    SD(pool.err);
    hipLaunchKernelGGL(k_iota, dim3(grid_for(m)), dim3(256), 0, s, iota, m);
    size_t tb = 0;
    SD(rocprim::radix_sort_pairs(nullptr, tb, key, skey, iota, st, (size_t)m, 0u, bits_for(n_keys), s));
    void *tmp = pool.get<char>(tb);
    SD(pool.err);
    SD(rocprim::radix_sort_pairs(tmp, tb, key, skey, iota, st, (size_t)m, 0u, bits_for(n_keys), s));
    SD(hipMemsetAsync(cnt, 0, ((size_t)n_keys + 1) * 4, s));
    hipLaunchKernelGGL(k_hist, dim3(grid_for(m)), dim3(256), 0, s, key, m, cnt);
    size_t tb2 = 0;
    SD(rocprim::exclusive_scan(nullptr, tb2, cnt, off, 0, (size_t)n_keys + 1, rocprim::plus<int32_t>(), s));
    void *tmp2 = pool.get<char>(tb2);
    SD(pool.err);
    SD(rocprim::exclusive_scan(tmp2, tb2, cnt, off, 0, (size_t)n_keys + 1, rocprim::plus<int32_t>(), s));
    return hipGetLastError();
}

struct Sorted {
    int32_t *skey = nullptr, *st = nullptr, *off = nullptr, *rank = nullptr;
};

template <int HMAX>
hipError_t launch_walk(const WalkArgs &a, unsigned blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_walk<HMAX>, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}
template <int HMAX>
int resident_blocks(int device) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_walk<HMAX>, 256, 0) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    return per_cu * cus;
}

// One walk over the first m tuples along `hub_item`'s side.  out = nullptr: count units / levels only.
struct WalkOut {
    int32_t *unit_of = nullptr, *unit_level = nullptr;
    uint8_t *pos = nullptr, *unit_len = nullptr;
};
hipError_t walk(DevPool &pool, int device, hipStream_t s, const Sorted &hubs, const Sorted &spokes, const int32_t *spoke_of, int64_t m, int32_t n_hub,
                int32_t n_spoke, int max_chain, const WalkOut *out, int64_t &n_units, int32_t &n_levels, bool &ok) {
    ok = false;
    int32_t *sp_h = pool.get<int32_t>((size_t)m), *want_h = pool.get<int32_t>((size_t)m), *sv = pool.get<int32_t>((size_t)n_hub * 4);
    unsigned long long *spoke = pool.get<unsigned long long>((size_t)n_spoke), *counters = pool.get<unsigned long long>(3);
    SD(pool.err);
    hipLaunchKernelGGL(k_lists, dim3(grid_for(m)), dim3(256), 0, s, hubs.st, spoke_of, spokes.rank, m, sp_h, want_h);
    hipLaunchKernelGGL(k_init_state, dim3(grid_for(n_hub)), dim3(256), 0, s, hubs.off, n_hub, sv);
    SD(hipMemsetAsync(spoke, 0, (size_t)n_spoke * 8, s));
    SD(hipMemsetAsync(counters, 0, 24, s));
    if (out) {
        SD(hipMemsetAsync(out->unit_level, 0, (size_t)m * 4, s));
    }
    WalkArgs a{hubs.st, sp_h, want_h, hubs.off, n_hub, max_chain, spoke, sv, out ? out->unit_of : nullptr, out ? out->pos : nullptr,
               out ? out->unit_level : nullptr, out ? out->unit_len : nullptr, counters, 4000};
    // a lane owns up to HMAX hub rows; the grid is what the device holds at once
    int hmax = 1;
    unsigned blocks = 0;
    for (; hmax <= 16; hmax *= 2) {
        const int rb = hmax == 1 ? resident_blocks<1>(device) : hmax == 2 ? resident_blocks<2>(device) : hmax == 4 ? resident_blocks<4>(device)
                       : hmax == 8 ? resident_blocks<8>(device) : resident_blocks<16>(device);
        if (rb <= 0) return hipErrorUnknown;
        // HALF of what the occupancy query reports: measured on MI355X, 1 954 blocks of 8 per CU reported were not all resident (the walk
        // then crawls through idle exits: 7.7 s for 50 M tuples), 1 221 were (12 ms)
        const int rb_used = std::max(1, rb / 2);
        if ((int64_t)rb_used * 256 * hmax >= n_hub) {
            blocks = (unsigned)std::min<int64_t>(rb_used, ((int64_t)n_hub + (int64_t)256 * hmax - 1) / ((int64_t)256 * hmax));
            break;
        }
    }
    if (getenv("CMI_SCHED_WALK_MEM")) blocks = 0; // tests: force the in-memory form
    if (!blocks) { // more hub rows than 16 per resident lane: the rows' state lives in memory
        hmax = 0;
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_walk_mem, 256, 0) != hipSuccess) return hipErrorUnknown;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return hipErrorUnknown;
        blocks = (unsigned)std::max(1, per_cu * cus / 2);
    }
    for (int launch = 0; launch < 4096; ++launch) {
        SD(hipMemsetAsync(counters + 2, 0, 8, s));
        switch (hmax) {
        case 0:
            hipLaunchKernelGGL(k_walk_mem, dim3(blocks), dim3(256), 0, s, a);
            SD(hipGetLastError());
            break;
        case 1: SD(launch_walk<1>(a, blocks, s)); break;
        case 2: SD(launch_walk<2>(a, blocks, s)); break;
        case 4: SD(launch_walk<4>(a, blocks, s)); break;
        case 8: SD(launch_walk<8>(a, blocks, s)); break;
        default: SD(launch_walk<16>(a, blocks, s)); break;
        }
        unsigned long long c[3];
        SD(hipMemcpyAsync(c, counters, 24, hipMemcpyDeviceToHost, s));
        SD(hipStreamSynchronize(s));
        if (c[2] == 0) {
            n_units = (int64_t)c[0];
            n_levels = (int32_t)c[1];
            ok = true;
            if (getenv("CMI_SETUP_TIMES"))
                fprintf(stderr, "[cmi] device schedule: walk over %lld tuples, %d hub rows (%s, %u blocks): %d launch(es), %lld units, %d levels\n", (long long)m,
                        n_hub, hmax ? "registers" : "memory", blocks, launch + 1, (long long)n_units, n_levels);
            return hipSuccess;
        }
    }
    return hipSuccess; // no convergence (should not happen): ok stays false, the host builder takes it
}

} // namespace

void ChainDeviceKeep::release() {
    for (int32_t **q : {&d_u, &d_j, &d_perm})
        if (*q) {
            (void)hipFree(*q);
            *q = nullptr;
        }
}

bool build_chain_schedule_device(int device, void *stream, int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int hub,
                                 int max_chain, ChainSchedule &out, ChainDeviceKeep *keep) {
    hipStream_t s = (hipStream_t)stream;
    out = ChainSchedule();
    out.unit_off.push_back(0);
    out.level_off.push_back(0);
    if (max_chain < 1) max_chain = 1;
    if (max_chain > 255) max_chain = 255;
    out.hub_is_item = hub == 0 ? 0 : 1;
    if (n <= 0) return true;
    if (n >= ((int64_t)1 << 31) - 1024) return false;
    if (hipSetDevice(device) != hipSuccess) return false;
    DevPool pool;
    int32_t *du = pool.get<int32_t>((size_t)n), *dj = pool.get<int32_t>((size_t)n);
    if (pool.err != hipSuccess) return false;
    if (hipMemcpyAsync(du, u, (size_t)n * 4, hipMemcpyHostToDevice, s) != hipSuccess) return false;
    if (hipMemcpyAsync(dj, j, (size_t)n * 4, hipMemcpyHostToDevice, s) != hipSuccess) return false;
    const bool times = getenv("CMI_SETUP_TIMES") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *w) {
        if (!times) return;
        (void)hipStreamSynchronize(s);
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[cmi] device schedule: %s %.3f s\n", w, std::chrono::duration<double>(t - T0).count());
        T0 = t;
    };
    auto fail = [&](const char *what, hipError_t e) {
        if (getenv("CMI_SETUP_TIMES")) fprintf(stderr, "[cmi] device schedule: %s: %s -- the host builder takes over\n", what, hipGetErrorString(e));
        (void)hipGetLastError();
        return false;
    };
    // both orders of the first m tuples (a stable sort of a prefix = the prefix's own sort): the rows' chains and the ranks along them
    auto sorted_sides = [&](int64_t m, Sorted &by_u, Sorted &by_j) -> hipError_t {
        SD(sort_by(pool, s, du, m, n_users, by_u.skey, by_u.st, by_u.off));
        SD(sort_by(pool, s, dj, m, n_items, by_j.skey, by_j.st, by_j.off));
        by_u.rank = pool.get<int32_t>((size_t)m);
        by_j.rank = pool.get<int32_t>((size_t)m);
        SD(pool.err);
        hipLaunchKernelGGL(k_rank, dim3(grid_for(m)), dim3(256), 0, s, by_u.skey, by_u.st, by_u.off, m, by_u.rank);
        hipLaunchKernelGGL(k_rank, dim3(grid_for(m)), dim3(256), 0, s, by_j.skey, by_j.st, by_j.off, m, by_j.rank);
        return hipGetLastError();
    };
    auto pick = [&](int64_t units_item, int64_t units_user) {
        if (hub == -2) return (double)units_user <= 1.3 * (double)units_item ? 0 : 1;
        if (hub == -3) return (double)units_item <= 1.3 * (double)units_user ? 1 : 0;
        return units_item <= units_user ? 1 : 0;
    };
    auto count_side = [&](int item_hub, int64_t m, const Sorted &by_u, const Sorted &by_j, int64_t &nu, bool &ok) -> hipError_t {
        int32_t nl = 0;
        return item_hub ? walk(pool, device, s, by_j, by_u, du, m, n_items, n_users, max_chain, nullptr, nu, nl, ok)
                        : walk(pool, device, s, by_u, by_j, dj, m, n_users, n_items, max_chain, nullptr, nu, nl, ok);
    };
    Sorted by_u, by_j;
    bool have_full = false;
    if (hub < 0 && n > ((int64_t)32 << 20)) { // large sets: the side is chosen on the first eighth of the tuples (build_chain_schedule)
        const int64_t m = n / 8;
        Sorted pu, pj;
        if (hipError_t e = sorted_sides(m, pu, pj)) return fail("prefix sorts", e);
        int64_t ui = 0, uu = 0;
        bool ok1 = false, ok0 = false;
        if (hipError_t e = count_side(1, m, pu, pj, ui, ok1)) return fail("prefix walk (items)", e);
        if (hipError_t e = count_side(0, m, pu, pj, uu, ok0)) return fail("prefix walk (users)", e);
        if (!ok1 || !ok0) return fail("prefix walk", hipSuccess);
        hub = pick(ui, uu);
        lap("side chosen on the first eighth");
    }
    if (hipError_t e = sorted_sides(n, by_u, by_j)) return fail("sorts", e);
    lap("upload + sorts");
    have_full = true;
    (void)have_full;
    if (hub < 0) {
        int64_t ui = 0, uu = 0;
        bool ok1 = false, ok0 = false;
        if (hipError_t e = count_side(1, n, by_u, by_j, ui, ok1)) return fail("count walk (items)", e);
        if (hipError_t e = count_side(0, n, by_u, by_j, uu, ok0)) return fail("count walk (users)", e);
        if (!ok1 || !ok0) return fail("count walk", hipSuccess);
        hub = pick(ui, uu);
    }
    out.hub_is_item = hub ? 1 : 0;
    WalkOut wo;
    wo.unit_of = pool.get<int32_t>((size_t)n);
    wo.unit_level = pool.get<int32_t>((size_t)n);
    wo.pos = pool.get<uint8_t>((size_t)n);
    wo.unit_len = pool.get<uint8_t>((size_t)n);
    if (pool.err != hipSuccess) return fail("allocation", pool.err);
    int64_t nu = 0;
    int32_t nl = 0;
    bool ok = false;
    {
        const hipError_t e = hub ? walk(pool, device, s, by_j, by_u, du, n, n_items, n_users, max_chain, &wo, nu, nl, ok)
                                 : walk(pool, device, s, by_u, by_j, dj, n, n_users, n_items, max_chain, &wo, nu, nl, ok);
        if (e != hipSuccess || !ok) return fail("walk", e);
    }
    lap("walk");
    // dense unit ids (CRS order of the first tuples), the units sorted by (level, length descending), offsets, the permutation
    int32_t *flag = pool.get<int32_t>((size_t)n), *dense = pool.get<int32_t>((size_t)n);
    uint32_t *key = pool.get<uint32_t>((size_t)nu), *skey = pool.get<uint32_t>((size_t)nu);
    int32_t *uid = pool.get<int32_t>((size_t)nu), *suid = pool.get<int32_t>((size_t)nu), *rank_of = pool.get<int32_t>((size_t)nu);
    int32_t *len_sorted = pool.get<int32_t>((size_t)nu + 1), *unit_off = pool.get<int32_t>((size_t)nu + 1), *perm = pool.get<int32_t>((size_t)n);
    uint8_t *ulen = pool.get<uint8_t>((size_t)nu);
    int64_t *level_off = pool.get<int64_t>((size_t)nl + 1);
    if (pool.err != hipSuccess) return fail("allocation", pool.err);
    hipLaunchKernelGGL(k_flags, dim3(grid_for(n)), dim3(256), 0, s, wo.unit_level, n, flag);
    {
        size_t tb = 0;
        if (hipError_t e = rocprim::exclusive_scan(nullptr, tb, flag, dense, 0, (size_t)n, rocprim::plus<int32_t>(), s)) return fail("scan", e);
        void *tmp = pool.get<char>(tb);
        if (pool.err != hipSuccess) return fail("allocation", pool.err);
        if (hipError_t e = rocprim::exclusive_scan(tmp, tb, flag, dense, 0, (size_t)n, rocprim::plus<int32_t>(), s)) return fail("scan", e);
    }
    hipLaunchKernelGGL(k_units, dim3(grid_for(n)), dim3(256), 0, s, wo.unit_level, wo.unit_len, dense, n, max_chain, key, uid, ulen);
    {
        const unsigned bits = bits_for((int64_t)nl * max_chain + 1);
        size_t tb = 0;
        if (hipError_t e = rocprim::radix_sort_pairs(nullptr, tb, key, skey, uid, suid, (size_t)nu, 0u, bits, s)) return fail("unit sort", e);
        void *tmp = pool.get<char>(tb);
        if (pool.err != hipSuccess) return fail("allocation", pool.err);
        if (hipError_t e = rocprim::radix_sort_pairs(tmp, tb, key, skey, uid, suid, (size_t)nu, 0u, bits, s)) return fail("unit sort", e);
    }
    hipLaunchKernelGGL(k_rank_units, dim3(grid_for(nu)), dim3(256), 0, s, suid, ulen, nu, rank_of, len_sorted);
    if (hipMemsetAsync(len_sorted + nu, 0, 4, s) != hipSuccess) return fail("memset", hipGetLastError());
    {
        size_t tb = 0;
        if (hipError_t e = rocprim::exclusive_scan(nullptr, tb, len_sorted, unit_off, 0, (size_t)nu + 1, rocprim::plus<int32_t>(), s)) return fail("scan", e);
        void *tmp = pool.get<char>(tb);
        if (pool.err != hipSuccess) return fail("allocation", pool.err);
        if (hipError_t e = rocprim::exclusive_scan(tmp, tb, len_sorted, unit_off, 0, (size_t)nu + 1, rocprim::plus<int32_t>(), s)) return fail("scan", e);
    }
    hipLaunchKernelGGL(k_perm, dim3(grid_for(n)), dim3(256), 0, s, wo.unit_of, wo.pos, dense, rank_of, unit_off, n, perm);
    hipLaunchKernelGGL(k_level_off, dim3((unsigned)((nl + 1 + 255) / 256)), dim3(256), 0, s, skey, nu, nl, max_chain, level_off);
    if (hipError_t e = hipGetLastError()) return fail("kernels", e);
    out.unit_off.assign((size_t)nu + 1, 0);
    out.level_off.assign((size_t)nl + 1, 0);
    if (keep) { // the caller takes the ids and the permutation over (they leave the pool)
        for (int32_t *q : {du, dj, perm}) pool.ptrs.erase(std::find(pool.ptrs.begin(), pool.ptrs.end(), (void *)q));
        keep->d_u = du;
        keep->d_j = dj;
        keep->d_perm = perm;
    } else {
        out.perm.resize((size_t)n);
        if (hipMemcpyAsync(out.perm.data(), perm, (size_t)n * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return fail("copy back", hipGetLastError());
    }
    if (hipMemcpyAsync(out.unit_off.data(), unit_off, ((size_t)nu + 1) * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return fail("copy back", hipGetLastError());
    if (hipMemcpyAsync(out.level_off.data(), level_off, ((size_t)nl + 1) * 8, hipMemcpyDeviceToHost, s) != hipSuccess) return fail("copy back", hipGetLastError());
    if (hipError_t e = hipStreamSynchronize(s)) return fail("synchronize", e);
    lap("unit sort + offsets + copy back");
    out.max_level_units = 0;
    for (int32_t l = 0; l < nl; ++l) out.max_level_units = std::max(out.max_level_units, out.level_off[(size_t)l + 1] - out.level_off[(size_t)l]);
    return true;
}

namespace {
template <typename R>
__global__ void k_stream(int64_t n, const int32_t *u, const int32_t *j, const int32_t *perm, const int32_t *ctx, const double *r, const int32_t *ctx_ptr,
                         const int32_t *ctx_conds, int dmax, int32_t *su, int32_t *sj, int32_t *sconds, R *sr) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        const int32_t t = perm[s];
        su[s] = u[t];
        sj[s] = j[t];
        sr[s] = (R)r[t];
        if (dmax > 0) {
            const int32_t c = ctx[t], b = ctx_ptr[c], e = ctx_ptr[c + 1];
            int32_t *row = sconds + s * dmax;
            for (int d = 0; d < dmax; ++d) row[d] = b + d < e ? ctx_conds[b + d] : -1;
        }
    }
}
} // namespace

int stream_build_device(void *stream, int64_t n, const int32_t *d_u, const int32_t *d_j, const int32_t *d_perm, const int32_t *ctx, const double *r,
                        const int32_t *d_ctx_ptr, const int32_t *d_ctx_conds, int dmax, bool f64, int32_t *d_su, int32_t *d_sj, int32_t *d_sconds,
                        void *d_sr) {
    hipStream_t s = (hipStream_t)stream;
    if (n <= 0) return (int)hipSuccess;
    DevPool pool;
    int32_t *d_ctx = dmax > 0 ? pool.get<int32_t>((size_t)n) : nullptr;
    double *d_r = pool.get<double>((size_t)n);
    if (pool.err != hipSuccess) return (int)pool.err;
    if (dmax > 0)
        if (hipError_t e = hipMemcpyAsync(d_ctx, ctx, (size_t)n * 4, hipMemcpyHostToDevice, s)) return (int)e;
    if (hipError_t e = hipMemcpyAsync(d_r, r, (size_t)n * 8, hipMemcpyHostToDevice, s)) return (int)e;
    if (f64) hipLaunchKernelGGL(k_stream<double>, dim3(grid_for(n)), dim3(256), 0, s, n, d_u, d_j, d_perm, d_ctx, d_r, d_ctx_ptr, d_ctx_conds, dmax, d_su, d_sj, d_sconds, (double *)d_sr);
    else hipLaunchKernelGGL(k_stream<float>, dim3(grid_for(n)), dim3(256), 0, s, n, d_u, d_j, d_perm, d_ctx, d_r, d_ctx_ptr, d_ctx_conds, dmax, d_su, d_sj, d_sconds, (float *)d_sr);
    if (hipError_t e = hipGetLastError()) return (int)e;
    return (int)hipStreamSynchronize(s); // the pool's buffers are freed on return
}

int arena_positions_device(void *stream, int64_t n, const int32_t *d_spoke_stream, int64_t n_spokes, int32_t *d_next, int32_t *d_first) {
    hipStream_t s = (hipStream_t)stream;
    if (n <= 0) return (int)hipSuccess;
    DevPool pool;
    int32_t *skey = nullptr, *spos = nullptr, *off = nullptr;
    if (hipError_t e = sort_by(pool, s, d_spoke_stream, n, (int32_t)n_spokes, skey, spos, off)) {
        if (getenv("CMI_SETUP_TIMES")) fprintf(stderr, "[cmi] device arena lists: %s -- the host walk takes over\n", hipGetErrorString(e));
        return (int)e;
    }
    hipLaunchKernelGGL(k_fill, dim3(grid_for(n_spokes)), dim3(256), 0, s, d_first, n_spokes, -1);
    hipLaunchKernelGGL(k_arena, dim3(grid_for(n)), dim3(256), 0, s, skey, spos, n, d_next, d_first);
    hipLaunchKernelGGL(k_arena_wrap, dim3(grid_for(n)), dim3(256), 0, s, skey, spos, n, d_next, d_first);
    if (hipError_t e = hipGetLastError()) return (int)e;
    return (int)hipStreamSynchronize(s); // the pool's buffers are freed on return
}

} // namespace cmi

// row_depth.hip -- how many spoke rows should a 16-lane group of the hub-chain kernel keep in flight?
// The chain kernel (chain_kernels.hip) reads row i+1 while it updates row i (depth 1).  The window experiment of row_bias.hip showed
// that this access pattern does not speed up when its rows sit in the L2 or the Infinity Cache -- so it is bound by requests in flight
// (Little's law), not by HBM bandwidth.  Here: the same read-modify-write of random 512-B rows (+ a scalar bias in its own table) with
// DEPTH = 1, 2, 3, 4 rows requested ahead, at 256 and 512 threads per workgroup.  No arithmetic.
// build: hipcc --offload-arch=gfx950 -O3 -o bin/row_depth row_depth.hip      run: bin/row_depth
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

struct Row { float4 a0, a1; float b; uint64_t r; };

template <int BIAS>
__device__ __forceinline__ Row load_row(const char *tab, const float *bias, uint64_t r, int l16) {
    Row x;
    x.r = r;
    x.a0 = *reinterpret_cast<const float4 *>(tab + r * 512 + 16 * l16);
    x.a1 = *reinterpret_cast<const float4 *>(tab + r * 512 + 256 + 16 * l16);
    x.b = 0.f;
    if (BIAS && l16 == 0) x.b = bias[r];
    return x;
}

template <int BIAS>
__device__ __forceinline__ void store_row(char *tab, float *bias, Row &x, int l16) {
    x.a0.x += 1.f; x.a1.w += 1.f;
    *reinterpret_cast<float4 *>(tab + x.r * 512 + 16 * l16) = x.a0;
    *reinterpret_cast<float4 *>(tab + x.r * 512 + 256 + 16 * l16) = x.a1;
    if (BIAS && l16 == 0) bias[x.r] = x.b + 1.f;
}

template <int DEPTH, int BIAS, int THREADS>
__global__ __launch_bounds__(THREADS) void rows(char *tab, float *bias, uint64_t n_rows, uint64_t n_groups, int per, uint64_t salt) {
    const uint64_t g = (uint64_t)blockIdx.x * (THREADS / 16) + (threadIdx.x >> 4);
    const int l16 = threadIdx.x & 15;
    if (g >= n_groups) return;
    auto row_of = [&](int i) { return mix((g * (uint64_t)per + (uint64_t)i) ^ salt) % n_rows; };
    Row ring[DEPTH + 1];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ring[d] = load_row<BIAS>(tab, bias, row_of(d < per ? d : per - 1), l16);
    // per is a multiple of DEPTH + 1: the ring is walked with static indices
    for (int i = 0; i < per; i += DEPTH + 1) {
#pragma unroll
        for (int s = 0; s <= DEPTH; ++s) {
            const int nxt = i + s + DEPTH;
            ring[(s + DEPTH) % (DEPTH + 1)] = load_row<BIAS>(tab, bias, row_of(nxt < per ? nxt : per - 1), l16);
            store_row<BIAS>(tab, bias, ring[s], l16);
        }
    }
}

template <int DEPTH, int BIAS, int THREADS>
static float run(char *tab, float *btab, uint64_t n_rows, int per) {
    const uint64_t n_groups = (uint64_t)(1u << 22) / per * 1;   // 4 M row visits per launch
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        const dim3 grid((unsigned)((n_groups + THREADS / 16 - 1) / (THREADS / 16)));
        hipEventRecord(e0);
        hipLaunchKernelGGL((rows<DEPTH, BIAS, THREADS>), grid, dim3(THREADS), 0, 0, tab, btab, n_rows, n_groups, per, (uint64_t)rep * 7919);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

template <int DEPTH, int BIAS>
static void report(char *tab, float *btab, uint64_t n_rows) {
    const int per = 12 * (DEPTH == 4 ? 5 : 1) / 1;   // multiple of DEPTH + 1 for DEPTH = 1, 2, 3 (12) and 4 (60)
    const double visits = (double)((uint64_t)(1u << 22) / per) * per;
    const float a = run<DEPTH, BIAS, 256>(tab, btab, n_rows, per), b = run<DEPTH, BIAS, 512>(tab, btab, n_rows, per);
    printf("{\"exp\": \"depth\", \"n_rows\": %llu, \"depth\": %d, \"bias\": %d, \"per\": %d, \"ms_256\": %.4f, \"row_GBps_256\": %.0f, \"ms_512\": %.4f, \"row_GBps_512\": %.0f}\n",
           (unsigned long long)n_rows, DEPTH, BIAS, per, a, visits * 1024.0 / a / 1e6, b, visits * 1024.0 / b / 1e6);
}

int main() {
    const size_t max_bytes = (size_t)6 << 30;
    char *tab; float *btab;
    if (hipMalloc(&tab, max_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&btab, (size_t)64 << 20);
    hipMemset(tab, 0, max_bytes);
    hipMemset(btab, 0, (size_t)64 << 20);
    for (uint64_t n_rows : {(uint64_t)1000000, (uint64_t)10000000}) {
        report<1, 0>(tab, btab, n_rows); report<2, 0>(tab, btab, n_rows); report<3, 0>(tab, btab, n_rows); report<4, 0>(tab, btab, n_rows);
        report<1, 1>(tab, btab, n_rows); report<2, 1>(tab, btab, n_rows); report<3, 1>(tab, btab, n_rows); report<4, 1>(tab, btab, n_rows);
    }
    return 0;
}

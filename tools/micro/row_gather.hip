// row_gather.hip -- what can this part sustain for the SGD kernels' access pattern?  16-lane groups read-modify-write random
// ROWS (R bytes, 16 B per lane per 256-B segment) of a table of S bytes: the pattern of P[u] / Q[j] traffic, with no arithmetic.
// Prints GB/s (read + write bytes) for a sweep of table sizes; `seq` visits rows in order (streaming reference).
// build: hipcc --offload-arch=gfx950 -O3 -o bin/row_gather row_gather.hip      run: bin/row_gather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
template <int NV, int MODE> // MODE 0: read+write, 1: read only, 2: write only
__global__ __launch_bounds__(256) void rows(float4 *tab, uint64_t n_rows, uint64_t n_groups, int per_group, int seq, float *sink, uint64_t salt) {
    const uint64_t g = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l16 = threadIdx.x & 15;
    if (g >= n_groups) return;
    float acc = 0.f;
    for (int i = 0; i < per_group; ++i) {
        const uint64_t id = g * per_group + i;
        const uint64_t r = seq ? id % n_rows : mix(id ^ salt) % n_rows;
        float4 *row = tab + r * (NV * 16) + l16;
        float4 v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = MODE == 2 ? make_float4(1.f, 2.f, 3.f, (float)i) : row[k * 16];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (MODE == 1) acc += v[k].x + v[k].w;
            else { v[k].x += 1.f; row[k * 16] = v[k]; }
        }
    }
    if (MODE == 1 && acc == 123.456f) *sink = acc;
}
template <int NV>
static void run(float4 *tab, size_t table_bytes, float *sink) {
    const uint64_t row_bytes = NV * 256, n_rows = table_bytes / row_bytes;
    const int per = 2;
    const uint64_t n_groups = 1u << 20; // 2M row visits per launch
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int seq = 0; seq < 2; ++seq) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                const dim3 grid((unsigned)((n_groups + 15) / 16));
                if (mode == 0) hipLaunchKernelGGL((rows<NV, 0>), grid, dim3(256), 0, 0, tab, n_rows, n_groups, per, seq, sink, (uint64_t)rep * 7919);
                if (mode == 1) hipLaunchKernelGGL((rows<NV, 1>), grid, dim3(256), 0, 0, tab, n_rows, n_groups, per, seq, sink, (uint64_t)rep * 7919);
                if (mode == 2) hipLaunchKernelGGL((rows<NV, 2>), grid, dim3(256), 0, 0, tab, n_rows, n_groups, per, seq, sink, (uint64_t)rep * 7919);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            const double bytes = (double)n_groups * per * row_bytes * (mode == 0 ? 2 : 1);
            printf("{\"row_bytes\": %llu, \"table_MB\": %zu, \"mode\": \"%s\", \"order\": \"%s\", \"GBps\": %.0f}\n", (unsigned long long)row_bytes,
                   table_bytes >> 20, mode == 0 ? "rw" : (mode == 1 ? "r" : "w"), seq ? "seq" : "random", bytes / best / 1e6);
        }
}
int main() {
    const size_t max_bytes = (size_t)12 << 30;
    float4 *tab; float *sink;
    if (hipMalloc(&tab, max_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 4);
    hipMemset(tab, 0, max_bytes);
    for (size_t mb : {64, 512, 2048, 12288}) {
        run<2>(tab, mb << 20, sink);  // 512-B rows (k=128 fp32)
        run<4>(tab, mb << 20, sink);  // 1-KB rows (k=256 fp32)
    }
    return 0;
}

// row_bias.hip -- round 3 questions about the hub-chain kernel's spoke-row traffic, with no arithmetic:
//  (1) what does the scalar bias of a spoke row cost when it lives in its own table (a 4-byte gather + scatter per row, the
//      product's layout in round 2) against riding in a padded tail of the row (stride 528 or 576 bytes)?
//  (2) what would ordering a launch's rows by page buy on a 5 GB table: rows drawn at random from the whole table against rows
//      drawn from a window of W bytes that advances with the workgroup index?
// 16-lane groups read-modify-write `per` rows of 512 B each (16 B per lane per 256-B segment), the next row's loads are issued
// before the current row's stores (the chain kernel's ping-pong).
// build: hipcc --offload-arch=gfx950 -O3 -o bin/row_bias row_bias.hip      run: bin/row_bias
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

// BIAS 0: none, 1: separate table (dword gather/scatter by lane 0), 2: in the row's tail (lane 0 moves one more dword of the same row)
template <int BIAS>
__global__ __launch_bounds__(256) void rows(char *tab, float *bias, uint64_t n_rows, uint64_t stride, uint64_t n_groups, int per,
                                            uint64_t window_rows, uint64_t salt) {
    const uint64_t g = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l16 = threadIdx.x & 15;
    if (g >= n_groups) return;
    // window mode: the rows of workgroup b come from [base, base + window_rows), base advancing with b over the table
    uint64_t base = 0, span = n_rows;
    if (window_rows && window_rows < n_rows) {
        span = window_rows;
        base = (uint64_t)((double)blockIdx.x / (double)gridDim.x * (double)(n_rows - window_rows));
    }
    auto row_of = [&](int i) { return base + mix((g * (uint64_t)per + (uint64_t)i) ^ salt) % span; };
    uint64_t r = row_of(0);
    float4 a0 = *reinterpret_cast<float4 *>(tab + r * stride + 16 * l16), a1 = *reinterpret_cast<float4 *>(tab + r * stride + 256 + 16 * l16);
    float b = 0.f;
    if (BIAS == 1 && l16 == 0) b = bias[r];
    if (BIAS == 2 && l16 == 0) b = *reinterpret_cast<float *>(tab + r * stride + 512);
    for (int i = 0; i < per; ++i) {
        uint64_t rn = r;
        float4 n0 = a0, n1 = a1;
        float nb = 0.f;
        if (i + 1 < per) {
            rn = row_of(i + 1);
            n0 = *reinterpret_cast<float4 *>(tab + rn * stride + 16 * l16);
            n1 = *reinterpret_cast<float4 *>(tab + rn * stride + 256 + 16 * l16);
            if (BIAS == 1 && l16 == 0) nb = bias[rn];
            if (BIAS == 2 && l16 == 0) nb = *reinterpret_cast<float *>(tab + rn * stride + 512);
        }
        a0.x += 1.f; a1.w += 1.f;
        *reinterpret_cast<float4 *>(tab + r * stride + 16 * l16) = a0;
        *reinterpret_cast<float4 *>(tab + r * stride + 256 + 16 * l16) = a1;
        if (BIAS == 1 && l16 == 0) bias[r] = b + 1.f;
        if (BIAS == 2 && l16 == 0) *reinterpret_cast<float *>(tab + r * stride + 512) = b + 1.f;
        r = rn; a0 = n0; a1 = n1; b = nb;
    }
}

// (3) "rows live where they are used next": every tuple owns an arena slot; a group READS its rows sequentially (stream order) and
// WRITES each updated row to a random slot (the slot of the row's next tuple).  bias = 1: the scalar rides in a padded 576-B slot? no --
// here it is still a separate gather/scatter table; bias = 0: rows only.
template <int BIAS>
__global__ __launch_bounds__(256) void rows_next_use(char *arena, float *bias, uint64_t n_slots, uint64_t n_bias, uint64_t n_groups, int per,
                                                     uint64_t salt) {
    const uint64_t g = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l16 = threadIdx.x & 15;
    if (g >= n_groups) return;
    const uint64_t first = (g * (uint64_t)per) % (n_slots - (uint64_t)per);
    float4 a0 = *reinterpret_cast<float4 *>(arena + first * 512 + 16 * l16), a1 = *reinterpret_cast<float4 *>(arena + first * 512 + 256 + 16 * l16);
    for (int i = 0; i < per; ++i) {
        float4 n0 = a0, n1 = a1;
        if (i + 1 < per) {
            n0 = *reinterpret_cast<float4 *>(arena + (first + i + 1) * 512 + 16 * l16);
            n1 = *reinterpret_cast<float4 *>(arena + (first + i + 1) * 512 + 256 + 16 * l16);
        }
        const uint64_t dst = mix((g * (uint64_t)per + (uint64_t)i) ^ salt) % n_slots;
        a0.x += 1.f; a1.w += 1.f;
        *reinterpret_cast<float4 *>(arena + dst * 512 + 16 * l16) = a0;
        *reinterpret_cast<float4 *>(arena + dst * 512 + 256 + 16 * l16) = a1;
        if (BIAS == 1 && l16 == 0) {
            const uint64_t b = mix(dst ^ 0x55) % n_bias;
            bias[b] += 1.f;
        }
        a0 = n0; a1 = n1;
    }
}

static float time_next_use(int bias, char *arena, float *btab, uint64_t n_slots, uint64_t n_bias) {
    const uint64_t n_groups = 1u << 20;
    const int per = 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        const dim3 grid((unsigned)((n_groups + 15) / 16));
        hipEventRecord(e0);
        if (bias == 0) hipLaunchKernelGGL((rows_next_use<0>), grid, dim3(256), 0, 0, arena, btab, n_slots, n_bias, n_groups, per, (uint64_t)rep * 7919);
        else hipLaunchKernelGGL((rows_next_use<1>), grid, dim3(256), 0, 0, arena, btab, n_slots, n_bias, n_groups, per, (uint64_t)rep * 7919);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

static float time_variant(int bias, char *tab, float *btab, uint64_t n_rows, uint64_t stride, uint64_t window_rows) {
    const uint64_t n_groups = 1u << 20; // 4 M row visits per launch
    const int per = 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        const dim3 grid((unsigned)((n_groups + 15) / 16));
        hipEventRecord(e0);
        if (bias == 0) hipLaunchKernelGGL((rows<0>), grid, dim3(256), 0, 0, tab, btab, n_rows, stride, n_groups, per, window_rows, (uint64_t)rep * 7919);
        if (bias == 1) hipLaunchKernelGGL((rows<1>), grid, dim3(256), 0, 0, tab, btab, n_rows, stride, n_groups, per, window_rows, (uint64_t)rep * 7919);
        if (bias == 2) hipLaunchKernelGGL((rows<2>), grid, dim3(256), 0, 0, tab, btab, n_rows, stride, n_groups, per, window_rows, (uint64_t)rep * 7919);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t max_bytes = (size_t)6 << 30;
    char *tab; float *btab;
    if (hipMalloc(&tab, max_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&btab, (size_t)64 << 20);
    hipMemset(tab, 0, max_bytes);
    hipMemset(btab, 0, (size_t)64 << 20);
    const double visits = (double)(1u << 20) * 4;
    for (uint64_t n_rows : {(uint64_t)1000000, (uint64_t)10000000}) {
        struct { const char *name; int bias; uint64_t stride; } v[] = {
            {"no bias, stride 512", 0, 512}, {"bias in its own table, stride 512", 1, 512}, {"bias in the row tail, stride 528", 2, 528},
            {"bias in the row tail, stride 576", 2, 576}, {"bias in the row tail, stride 640", 2, 640}, {"no bias, stride 528", 0, 528}};
        for (auto &x : v) {
            const float ms = time_variant(x.bias, tab, btab, n_rows, x.stride, 0);
            printf("{\"exp\": \"bias\", \"n_rows\": %llu, \"variant\": \"%s\", \"ms\": %.4f, \"Mrows_per_s\": %.0f, \"row_GBps\": %.0f}\n",
                   (unsigned long long)n_rows, x.name, ms, visits / ms / 1e3, visits * 1024.0 / ms / 1e6);
        }
    }
    // (2) page locality on the 10 M-row (5 GB) table: window of W MB advancing with the workgroup index
    for (uint64_t wmb : {(uint64_t)0, (uint64_t)2048, (uint64_t)512, (uint64_t)128, (uint64_t)32, (uint64_t)8, (uint64_t)2}) {
        const uint64_t wrows = wmb ? (wmb << 20) / 512 : 0;
        for (int bias : {0, 1}) {
            const float ms = time_variant(bias, tab, btab, 10000000, 512, wrows);
            printf("{\"exp\": \"window\", \"n_rows\": 10000000, \"window_MB\": %llu, \"bias\": %d, \"ms\": %.4f, \"row_GBps\": %.0f}\n",
                   (unsigned long long)wmb, bias, ms, visits * 1024.0 / ms / 1e6);
        }
    }
    // (3) sequential reads + random writes over arenas of 25.6 GB (C3: one slot per tuple) -- allocated separately
    hipFree(tab);
    for (uint64_t gb : {(uint64_t)2, (uint64_t)25, (uint64_t)100}) {
        const size_t bytes = (size_t)gb << 30;
        char *arena;
        if (hipMalloc(&arena, bytes) != hipSuccess) { printf("{\"exp\": \"next_use\", \"error\": \"alloc %llu GB failed\"}\n", (unsigned long long)gb); continue; }
        hipMemset(arena, 0, bytes);
        for (int bias : {0, 1}) {
            const float ms = time_next_use(bias, arena, btab, bytes / 512, 1000000);
            printf("{\"exp\": \"next_use\", \"arena_GB\": %llu, \"bias\": %d, \"ms\": %.4f, \"row_GBps\": %.0f}\n", (unsigned long long)gb, bias, ms,
                   visits * 1024.0 / ms / 1e6);
        }
        hipFree(arena);
    }
    return 0;
}

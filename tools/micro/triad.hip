// HBM calibration: device-to-device copy and triad on the box, next to the 8 TB/s spec peak.
// hipcc --offload-arch=gfx950 -O3 triad.hip -o triad
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ a, float4 *__restrict__ c, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c[i] = a[i];
}
__global__ __launch_bounds__(256) void k_triad(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ c, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i];
        c[i] = make_float4(x.x + 3.f * y.x, x.y + 3.f * y.y, x.z + 3.f * y.z, x.w + 3.f * y.w);
    }
}
int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    float4 *a, *b, *c;
    (void)hipMalloc(&a, bytes);
    (void)hipMalloc(&b, bytes);
    (void)hipMalloc(&c, bytes);
    (void)hipMemset(a, 0, bytes);
    (void)hipMemset(b, 0, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms;
    for (int blocks : {4096, 16384, 65536}) {
        hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, c, n);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, c, n);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("copy  blocks=%6d: %.0f GB/s (read+write)\n", blocks, 2.0 * bytes * 10 / (ms * 1e-3) / 1e9);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_triad, dim3(blocks), dim3(256), 0, 0, a, b, c, n);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("triad blocks=%6d: %.0f GB/s (2 reads + 1 write)\n", blocks, 3.0 * bytes * 10 / (ms * 1e-3) / 1e9);
    }
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) (void)hipMemcpyAsync(c, a, bytes, hipMemcpyDeviceToDevice, 0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpy D2D: %.0f GB/s (read+write)\n", 2.0 * bytes * 10 / (ms * 1e-3) / 1e9);
    return 0;
}

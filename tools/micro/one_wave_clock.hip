// one_wave_clock.hip -- what clock does a kernel of ONE wave run at, and what does a dependent VALU / readlane / LDS link cost there?
// (CAMF_C is a single dependency chain: one wave, 0.3-0.6 us per link.)  s_memtime counts shader clocks, s_memrealtime 100 MHz.
// build: hipcc --offload-arch=gfx950 -O3 -o bin/one_wave_clock one_wave_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void chain(float *out, long long *t, int n, int mode) {
    __shared__ float lds[64];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float x = out[threadIdx.x];
    const long long c0 = clock64(), w0 = wall_clock64();
    if (mode == 0) {
        for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f;                       // 2 dependent VALU ops per link (no fma contraction)
    } else if (mode == 1) {
        for (int i = 0; i < n; ++i) {                                               // VALU + readlane (VGPR -> SGPR -> VALU)
            const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), i & 63));
            x = x * 1.0001f + s;
        }
    } else if (mode == 2) {
        for (int i = 0; i < n; ++i) {                                               // LDS round trip per link
            lds[threadIdx.x] = x;
            x = lds[(threadIdx.x + 1) & 63] * 1.0001f + 0.5f;
        }
    } else {
        for (int i = 0; i < n; ++i) {                                               // 6-step DPP reduction + readlane per link
            float y = x;
            for (int m = 32; m >= 1; m >>= 1) y += __shfl_xor(y, m, 64);
            x = x * 0.5f + y * 1e-3f;
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}

__global__ void busy(float *out, int n) {   // keeps the other CUs occupied
    float x = out[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main() {
    float *out, *out2; long long *t, h[2];
    hipMalloc(&out, 1 << 22); hipMalloc(&out2, 1 << 24); hipMalloc(&t, 16);
    hipMemset(out, 0, 1 << 22); hipMemset(out2, 0, 1 << 24);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    const int n = 2000000;
    const char *names[] = {"2 dependent VALU ops", "readlane + 2 VALU", "LDS write + read + 2 VALU", "64-lane shuffle reduction + 2 VALU"};
    for (int with_busy = 0; with_busy < 2; ++with_busy)
        for (int mode = 0; mode < 4; ++mode) {
            if (with_busy) hipLaunchKernelGGL(busy, dim3(4096), dim3(256), 0, s2, out2, 400000);
            hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, out, t, n, mode);
            hipStreamSynchronize(s1);
            hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            hipDeviceSynchronize();
            const double secs = (double)h[1] / 100e6;
            printf("{\"other_CUs_busy\": %d, \"link\": \"%s\", \"shader_MHz\": %.0f, \"cycles_per_link\": %.1f, \"ns_per_link\": %.1f}\n", with_busy,
                   names[mode], (double)h[0] / secs / 1e6, (double)h[0] / n, secs / n * 1e9);
        }
    return 0;
}

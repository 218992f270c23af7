// barrier_cost.hip -- what a workgroup barrier costs a 1024-thread (16-wave) workgroup that has a CU to itself, and what LDS hand-offs
// between its waves cost: the SVD++ team kernel (svdpp_team.hip) pays three barriers per chain link.
// build: hipcc --offload-arch=gfx950 -O3 -o bin/barrier_cost barrier_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, long long *t, int n) {
    __shared__ float s[1024];
    float x = threadIdx.x;
    s[threadIdx.x] = x;
    __syncthreads();
    const long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (MODE == 1) __syncthreads();
        if (MODE == 2) { // LDS write, barrier, LDS read of another wave's value
            s[threadIdx.x] = x;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            x = s[(threadIdx.x + 64) & 1023] + 1.f;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    const long long c1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) t[0] = c1 - c0;
}

int main() {
    float *out; long long *t, h;
    hipMalloc(&out, 4096); hipMalloc(&t, 8);
    const int n = 200000;
    const char *names[] = {"s_waitcnt lgkmcnt(0) + s_barrier", "__syncthreads()", "LDS write + barrier + LDS read + barrier"};
    for (int threads : {1024, 256, 64})
        for (int mode = 0; mode < 3; ++mode) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, out, t, n);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, out, t, n);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, out, t, n);
            hipDeviceSynchronize();
            hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
            printf("{\"threads\": %d, \"step\": \"%s\", \"cycles_per_step\": %.1f}\n", threads, names[mode], (double)h / n);
        }
    return 0;
}

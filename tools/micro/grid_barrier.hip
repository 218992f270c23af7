// microbenchmark: cost of an in-kernel grid-wide barrier (one persistent launch instead of one launch per level)
// hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_barrier(unsigned *counter, int rounds, float *sink, const float *src, int work) {
    const unsigned nb = gridDim.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        for (int w = 0; w < work; ++w) acc += src[(size_t)((blockIdx.x * 256 + threadIdx.x) + (size_t)w * nb * 256 + (size_t)r * 1024) % (1u << 24)];
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE); // agent scope by default in HIP device code? use explicit scope below
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * nb;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (acc == 1.2345f) sink[0] = acc;
}

__global__ void k_empty(float *sink) {
    if (sink == nullptr) sink[0] = 1.f;
}

int main() {
    unsigned *counter;
    float *sink, *src;
    hipMalloc(&counter, 4);
    hipMalloc(&sink, 4);
    hipMalloc(&src, (size_t)(1u << 24) * 4);
    hipMemset(src, 0, (size_t)(1u << 24) * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int rounds = 1000;
    for (int work : {0, 4}) {
        for (int blocks : {4, 8, 16, 32, 64, 256, 512, 1024, 2048}) {
            hipMemset(counter, 0, 4);
            void *args[] = {&counter, (void *)&rounds, &sink, &src, (void *)&work};
            hipEventRecord(e0);
            hipError_t e = hipLaunchCooperativeKernel((void *)k_barrier, dim3(blocks), dim3(256), args, 0, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("work=%d blocks=%5d: %s %.3f us per barrier round\n", work, blocks, hipGetErrorString(e), ms * 1e3 / rounds);
        }
    }
    // reference: a chain of empty kernels on one stream, and the same chain in a graph
    hipStream_t s;
    hipStreamCreate(&s);
    hipEventRecord(e0, s);
    for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_empty, dim3(3300), dim3(256), 0, s, sink);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("empty-kernel chain (3300 blocks), eager: %.3f us per launch\n", ms * 1e3 / rounds);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_empty, dim3(3300), dim3(256), 0, s, sink);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("empty-kernel chain (3300 blocks), hipGraph: %.3f us per launch\n", ms * 1e3 / rounds);
    return 0;
}

// What one lone wavefront per SIMD can issue on gfx950: cycles per instruction for (a) a dependent v_fma chain, (b) four independent chains
// interleaved, (c) a dependent v_add_f32_dpp chain, (d) dependent VALU alternating with independent SALU, (e) a dependent packed chain.  Sizes the owner kernel's step
// (DESIGN.md section 5): its hottest owner is one wavefront running a ~90-instruction dependent step.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/issue_rate.hip -o tools/micro/bin/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256, 1) void k(float *out, long long *cyc, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f;
    int s = iters;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p2 = {a, c}, q2 = {b, b};
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(b));) }
        if (KIND == 1) {
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3"
                               : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));)
        }
        if (KIND == 2) { REP16(asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a));) }
        if (KIND == 3) { REP16(asm volatile("v_fma_f32 %0, %0, %2, %0\n s_mul_i32 %1, %1, 3" : "+v"(a), "+s"(s) : "v"(b));) }
        if (KIND == 4) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p2) : "v"(q2));) }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = a + c + d + e + (float)s + p2.x + p2.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int per_iter) {
    float *out;
    long long *cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, cyc, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * per_iter;
    printf("{\"kind\": \"%s\", \"instructions\": %.0f, \"ns_per_instruction\": %.3f, \"clock64_ticks_per_instruction\": %.3f}\n", name, n,
           ms * 1e6 / n, (double)h[0] / n);
    fflush(stdout);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("dependent v_fma_f32 chain", 16);
    run<1>("four independent v_fma_f32 chains interleaved", 64);
    run<2>("dependent v_add_f32_dpp chain (each with its s_nop 1)", 32);
    run<3>("dependent v_fma_f32 alternating with s_mul_i32", 32);
    run<4>("dependent v_pk_fma_f32 chain", 16);
    return 0;
}

// alloc_time.hip -- how long does the device take to hand out a 100-GB block?  (north_star's spoke arena: cmi_set_ratings is bound by it)
// One method per PROCESS (a freed block is handed back from a cache by the next allocation of the same size, which hides the cost):
//   alloc_time <GB> <method>   0 hipMalloc, 1 hipMallocAsync, 2 / 3 / 4 hipMemCreate + hipMemMap in 1-GB / 8-GB / one chunk(s), 5 hipMalloc in 1-GB pieces
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/alloc_time.hip -o tools/micro/bin/alloc_time
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char *p, size_t n, size_t stride, unsigned long long *cnt) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (i < n) {
        p[i] = 1;
        if ((i / stride) % 4096 == 0) atomicAdd(cnt, 1ull);
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv) {
    const size_t gb = argc > 1 ? (size_t)atol(argv[1]) : 100;
    const int method = argc > 2 ? atoi(argv[2]) : 0;
    const size_t bytes = gb << 30;
    setvbuf(stdout, nullptr, _IONBF, 0);
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    hipStream_t s; CK(hipStreamCreate(&s));
    unsigned long long *cnt; CK(hipMalloc(&cnt, 8)); CK(hipMemset(cnt, 0, 8));
    double a = 0, ft = 0;
    auto first_touch = [&](char *p, size_t len) -> double {
        double t = now();
        const size_t stride = 4096, nth = len / stride;
        hipLaunchKernelGGL(touch, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, s, p, len, stride, cnt);
        if (hipStreamSynchronize(s) != hipSuccess) printf("touch failed\n");
        return now() - t;
    };
    const char *what = "";
    if (method == 0) {
        what = "hipMalloc";
        void *p = nullptr; double t = now(); CK(hipMalloc(&p, bytes)); a = now() - t; ft = first_touch((char *)p, bytes);
    } else if (method == 1) {
        what = "hipMallocAsync";
        void *p = nullptr; double t = now(); CK(hipMallocAsync(&p, bytes, s)); CK(hipStreamSynchronize(s)); a = now() - t; ft = first_touch((char *)p, bytes);
    } else if (method == 5) {
        what = "hipMalloc in 1-GB pieces";
        double t = now();
        std::vector<void *> ps(gb);
        for (size_t i = 0; i < gb; ++i) CK(hipMalloc(&ps[i], (size_t)1 << 30));
        a = now() - t;
        for (size_t i = 0; i < gb; ++i) ft += first_touch((char *)ps[i], (size_t)1 << 30);
    } else {
        const size_t chunk_gb = method == 2 ? 1 : method == 3 ? 8 : gb;
        what = method == 2 ? "hipMemCreate + hipMemMap, 1-GB chunks" : method == 3 ? "hipMemCreate + hipMemMap, 8-GB chunks" : "hipMemCreate + hipMemMap, one chunk";
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        const size_t chunk = chunk_gb << 30, nchunk = (bytes + chunk - 1) / chunk, total = nchunk * chunk;
        double t = now();
        void *va = nullptr; CK(hipMemAddressReserve(&va, total, gran, nullptr, 0));
        hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
        double tc = 0, tm = 0, ta = 0;
        for (size_t i = 0; i < nchunk; ++i) {
            hipMemGenericAllocationHandle_t h;
            double t0 = now(); CK(hipMemCreate(&h, chunk, &prop, 0));
            double t1 = now(); CK(hipMemMap((char *)va + i * chunk, chunk, 0, h, 0));
            double t2 = now(); CK(hipMemSetAccess((char *)va + i * chunk, chunk, &ad, 1));
            double t3 = now();
            tc += t1 - t0; tm += t2 - t1; ta += t3 - t2;
        }
        a = now() - t;
        printf("  granularity %zu: create %.3f s, map %.3f s, set access %.3f s\n", gran, tc, tm, ta);
        ft = first_touch((char *)va, total);
    }
    unsigned long long c = 0; CK(hipMemcpy(&c, cnt, 8, hipMemcpyDeviceToHost));
    printf("{\"what\": \"%s\", \"GB\": %zu, \"alloc_s\": %.3f, \"first_touch_s\": %.3f, \"pages_touched\": %llu}\n", what, gb, a, ft, c * 4096ull);
    return 0;
}

// microbenchmark: 16-byte random gathers over a 400 MB array, unrestricted vs window by window (windows of 25/50/100 MB):
// does restricting concurrent gathers to a cache-sized window (Infinity Cache 256 MB) raise the gather rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ void k_gather(const int *__restrict__ sup, const double2 *__restrict__ R, double *out, long n) {
    double s = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double2 r = R[sup[i]];
        s += r.x * r.y;
    }
    if (s == 1.2345) out[0] = s;
}

int main() {
    const long n = 25000000;
    std::vector<int> idx(n);
    srand(3);
    for (long i = 0; i < n; ++i) idx[i] = (int)((((long)rand() << 15) ^ rand()) % n);
    int *ds;
    double2 *dR;
    double *dout;
    (void)hipMalloc(&ds, n * 4);
    (void)hipMalloc(&dR, n * 16);
    (void)hipMalloc(&dout, 8);
    (void)hipMemset(dR, 0, n * 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int windows : {1, 4, 8, 16}) {
        std::vector<int> s = idx;
        const long wsz = (n + windows - 1) / windows;
        // stable partition by window: the index stream visits window 0 first, then window 1, ... (random inside a window)
        std::stable_sort(s.begin(), s.end(), [&](int a, int b) { return a / wsz < b / wsz; });
        (void)hipMemcpy(ds, s.data(), n * 4, hipMemcpyHostToDevice);
        std::vector<long> cnts(windows, 0);
        for (long i = 0; i < n; ++i) cnts[s[i] / wsz]++;
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            // one launch per window keeps the windows apart in time
            long off = 0;
            for (int w = 0; w < windows; ++w) {
                hipLaunchKernelGGL(k_gather, dim3(4096), dim3(256), 0, 0, ds + off, dR, dout, cnts[w]);
                off += cnts[w];
            }
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("windows=%2d (%.0f MB each): %.3f ms  %.1f G gathers/s\n", windows, wsz * 16 / 1e6, best, n / best / 1e6);
    }
    return 0;
}

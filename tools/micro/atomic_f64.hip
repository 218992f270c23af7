// microbenchmark: fp64 atomic accumulation of (num, den) into a small per-coordinate table from a rating stream
// (the FM item-field reduce).  hipcc --offload-arch=gfx950 -O3 atomic_f64.hip -o atomic_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_atomic(const int *__restrict__ j, const double *__restrict__ err, double *num, double *den, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double e = err[i];
        const int c = j[i];
        unsafeAtomicAdd(num + c, e * 0.5);
        unsafeAtomicAdd(den + c, 0.25);
    }
}
__global__ void k_atomic2(const int *__restrict__ j, const double *__restrict__ err, double2 *acc, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double e = err[i];
        const int c = j[i];
        unsafeAtomicAdd(&acc[c].x, e * 0.5);
        unsafeAtomicAdd(&acc[c].y, 0.25);
    }
}
__global__ void k_stream(const int *__restrict__ j, double *err, const double *__restrict__ delta, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        err[i] = err[i] + delta[j[i]];
}
__global__ void k_gather(const int *__restrict__ sup, const double *__restrict__ err, double *out, long n) {
    double s = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += err[sup[i]];
    if (s == 1.2345) out[0] = s;
}

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 25000000;
    const int m = argc > 2 ? atoi(argv[2]) : 500000;
    std::vector<int> hj(n), hs(n);
    srand(1);
    for (long i = 0; i < n; ++i) {
        hj[i] = (int)((((long)rand() << 15) ^ rand()) % m);
        hs[i] = (int)((((long)rand() << 15) ^ rand()) % n);
    }
    int *dj, *ds;
    double *derr, *dnum, *dden, *ddelta;
    double2 *dacc;
    hipMalloc(&dj, n * 4);
    hipMalloc(&ds, n * 4);
    hipMalloc(&derr, n * 8);
    hipMalloc(&dnum, m * 8);
    hipMalloc(&dden, m * 8);
    hipMalloc(&ddelta, m * 8);
    hipMalloc(&dacc, m * 16);
    hipMemcpy(dj, hj.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(ds, hs.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(derr, 0, n * 8);
    hipMemset(dnum, 0, m * 8);
    hipMemset(dden, 0, m * 8);
    hipMemset(ddelta, 0, m * 8);
    hipMemset(dacc, 0, m * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](const char *name, auto f) {
        f();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %.3f ms/pass  (%.1f G elem/s)\n", name, ms / 5, n / (ms / 5) / 1e6);
    };
    for (int blocks : {2048, 8192}) {
        printf("blocks=%d\n", blocks);
        time("atomic num+den (2 arrays)", [&] { hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, dj, derr, dnum, dden, n); });
        time("atomic num+den (double2)", [&] { hipLaunchKernelGGL(k_atomic2, dim3(blocks), dim3(256), 0, 0, dj, derr, dacc, n); });
        time("stream apply err+=delta[j]", [&] { hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, dj, derr, ddelta, n); });
        time("random gather err[sup[i]]", [&] { hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, ds, derr, dnum, n); });
    }
    return 0;
}

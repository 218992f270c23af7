// calib_kernels.hip -- what does THIS part sustain?  Two calibration kernels bench.py runs beside the SGD epoch so that its
// roofline line can quote a measured ceiling next to the 8 TB/s spec peak (MI355X_MICROARCH.md: ~6.3 TB/s float4 copy):
//   copy     streaming float4 copy, 16 B per lane (bytes = read + write)
//   row_rw   16-lane groups read-modify-write RANDOM 512-byte rows of a large table -- the access pattern of the P[u] / Q[j]
//            traffic of the SGD kernels, with no arithmetic and no dependent id loads (tools/micro/row_gather.hip is the sweep)
#include "../../include/carskit_mi355x.h"

#include <hip/hip_runtime.h>

namespace {

// one float4 per thread, blocks in address order: the shape that reaches the part's copy rate (tools/micro/copy_variants.hip:
// 6.24 TB/s, against 4.5-5.1 TB/s for grid-stride loops of any unroll and 4.86 TB/s for hipMemcpy D2D on the same box)
__global__ __launch_bounds__(256) void calib_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}

__device__ __forceinline__ uint64_t calib_mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void calib_row_rw(float4 *tab, uint64_t n_rows, uint64_t n_groups, uint64_t salt) {
    const uint64_t g = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (g >= n_groups) return;
    const int l16 = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float4 *row = tab + (calib_mix((g * 2 + i) ^ salt) % n_rows) * 32 + l16; // 512-byte rows = 2 x 256-byte segments
        float4 a = row[0], b = row[16];
        a.x += 1.f;
        b.x += 1.f;
        row[0] = a;
        row[16] = b;
    }
}

} // namespace

extern "C" int cmi_measure_hbm(int device, int64_t bytes, double out[2]) {
    if (!out || bytes < (1 << 20)) return CMI_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CMI_E_NO_DEVICE;
    if (device < 0 || device >= ndev) return CMI_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return CMI_E_HIP;
    float4 *buf = nullptr;
    if (hipMalloc((void **)&buf, (size_t)bytes) != hipSuccess) return CMI_E_HIP;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = CMI_OK;
    if (hipMemset(buf, 0, (size_t)bytes) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = CMI_E_HIP;
    const int64_t n4 = bytes / 16 / 2; // copy the first half onto the second
    const uint64_t n_rows = (uint64_t)bytes / 512, n_groups = (uint64_t)1 << 21;
    double best[2] = {0.0, 0.0};
    for (int rep = 0; rep < 4 && rc == CMI_OK; ++rep)
        for (int which = 0; which < 2; ++which) {
            (void)hipEventRecord(e0, nullptr);
            if (which == 0) hipLaunchKernelGGL(calib_copy, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, nullptr, buf, buf + n4, n4);
            else hipLaunchKernelGGL(calib_row_rw, dim3((unsigned)(n_groups / 16)), dim3(256), 0, nullptr, buf, n_rows, n_groups, (uint64_t)rep * 7919);
            (void)hipEventRecord(e1, nullptr);
            float ms = 0.f;
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) {
                rc = CMI_E_HIP;
                break;
            }
            const double moved = which == 0 ? (double)n4 * 32.0 : (double)n_groups * 2 * 512 * 2;
            const double gbps = moved / (ms * 1e-3) / 1e9;
            if (rep > 0 && gbps > best[which]) best[which] = gbps; // rep 0 warms up
        }
    out[0] = best[0];
    out[1] = best[1];
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    return rc;
}

// store_data_hazard.hip -- does a 128-bit buffer store whose soffset is an SGPR need wait states before a VALU write of its data
// registers on gfx950?  (Round 6: the cause of the owner epoch's rare wrong low words beside another owner epoch.)
//
// The gfx9-family rule: "VMEM store of more than 64 bits of data, followed by a VALU write of the VGPRs holding the write data:
// 1 wait state (2 on gfx940+)" -- the store reads its data after it has issued.  The ISA manuals exempt buffer stores that "use an
// SGPR for the offset", and LLVM's hazard recognizer follows them (GCNHazardRecognizer::createsVALUHazard returns no hazard when the
// MUBUF soffset operand is a register), so for `buffer_store_dwordx4 v[a:a+3], voff, rsrc, sN offen` the compiler inserts nothing.
// This program issues exactly that store and overwrites two of its data registers with a poison value W wait states later
// (W = 0, 1, 2, 3), from many wavefronts per SIMD so that the memory pipeline is backed up, each store to its own 1-KB slot; a
// second kernel counts the slots in which the poison reached memory.
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_data_hazard.hip -o tools/micro/bin/store_data_hazard
//   tools/micro/bin/store_data_hazard [waves per SIMD = 8] [stores per wave = 256]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                                        \
    do {                                                                                                \
        hipError_t e_ = (x);                                                                            \
        if (e_ != hipSuccess) {                                                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                     \
            exit(1);                                                                                    \
        }                                                                                               \
    } while (0)

static const uint32_t POISON = 0xDEADBEEFu;

// W = wait states between the store and the first VALU write of its data registers; SGPR_SOFF: the slot offset travels in soffset (the
// exempted form) or is added into voffset with soffset = 0 (the form the compiler itself protects)
template <int W, bool SGPR_SOFF>
__global__ __launch_bounds__(256) void hazard(uint32_t *buf, int iters, uint32_t poison) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const uint64_t base = (uint64_t)buf;
    u32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((uint32_t)base);
    rs.y = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32) & 0xffffu);
    rs.z = 0xffffffffu;
    rs.w = 0x00020000u;
    for (int i = 0; i < iters; ++i) {
        const uint32_t slot = (uint32_t)wave * (uint32_t)iters + (uint32_t)i;
        const uint32_t lo = slot * 64u + (uint32_t)lane, hi = ~lo & 0x7fffffffu, tag = (uint32_t)i + 1u;
        const uint32_t soff = SGPR_SOFF ? slot * 1024u : 0u;
        const uint32_t voff = (uint32_t)lane * 16u + (SGPR_SOFF ? 0u : slot * 1024u);
        if constexpr (SGPR_SOFF) {
            asm volatile("v_mov_b32 v100, %0\n\t"
                         "v_mov_b32 v101, %2\n\t"
                         "v_mov_b32 v102, %1\n\t"
                         "v_mov_b32 v103, %2\n\t"
                         "s_nop 4\n\t"
                         "buffer_store_dwordx4 v[100:103], %3, %4, %5 offen sc0 sc1\n\t"
                         ".rept %7\n\t"
                         "s_nop 0\n\t"
                         ".endr\n\t"
                         "v_mov_b32 v100, %6\n\t"
                         "v_mov_b32 v102, %6\n\t"
                         :
                         : "v"(lo), "v"(hi), "v"(tag), "v"(voff), "s"(rs), "s"(soff), "v"(poison), "n"(W)
                         : "v100", "v101", "v102", "v103", "memory");
        } else {
            asm volatile("v_mov_b32 v100, %0\n\t"
                         "v_mov_b32 v101, %2\n\t"
                         "v_mov_b32 v102, %1\n\t"
                         "v_mov_b32 v103, %2\n\t"
                         "s_nop 4\n\t"
                         "buffer_store_dwordx4 v[100:103], %3, %4, 0 offen sc0 sc1\n\t"
                         ".rept %6\n\t"
                         "s_nop 0\n\t"
                         ".endr\n\t"
                         "v_mov_b32 v100, %5\n\t"
                         "v_mov_b32 v102, %5\n\t"
                         :
                         : "v"(lo), "v"(hi), "v"(tag), "v"(voff), "s"(rs), "v"(poison), "n"(W)
                         : "v100", "v101", "v102", "v103", "memory");
        }
    }
}

// per 16-byte piece {lo, tag, hi, tag}: poisoned (either data word), correct, or something else
__global__ void audit(const uint32_t *buf, int64_t pieces, int iters, uint32_t poison, unsigned long long *out) {
    unsigned long long bad_lo = 0, bad_hi = 0, good = 0, other = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < pieces; p += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t *w = buf + p * 4;
        const uint32_t slot = (uint32_t)(p / 64), lane = (uint32_t)(p % 64);
        const uint32_t lo = slot * 64u + lane, hi = ~lo & 0x7fffffffu, tag = slot % (uint32_t)iters + 1u;
        const bool tags = w[1] == tag && w[3] == tag;
        if (tags && w[0] == lo && w[2] == hi) ++good;
        else {
            if (tags && w[0] == poison) ++bad_lo;
            if (tags && w[2] == poison) ++bad_hi;
            if (!(tags && (w[0] == poison || w[2] == poison))) ++other;
        }
    }
    atomicAdd(out + 0, bad_lo);
    atomicAdd(out + 1, bad_hi);
    atomicAdd(out + 2, good);
    atomicAdd(out + 3, other);
}

template <int W, bool SGPR_SOFF>
static void run(uint32_t *buf, int waves, int iters, unsigned long long *d_out) {
    const int64_t pieces = (int64_t)waves * iters * 64;
    CHECK(hipMemset(buf, 0, (size_t)pieces * 16));
    CHECK(hipMemset(d_out, 0, 32));
    hipLaunchKernelGGL((hazard<W, SGPR_SOFF>), dim3(waves / 4), dim3(256), 0, 0, buf, iters, POISON);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(audit, dim3(4096), dim3(256), 0, 0, buf, pieces, iters, POISON, d_out);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[4];
    CHECK(hipMemcpy(h, d_out, 32, hipMemcpyDeviceToHost));
    printf("{\"soffset\": \"%s\", \"wait_states\": %d, \"stores\": %lld, \"pieces\": %lld, \"poisoned_lo\": %llu, \"poisoned_hi\": %llu, \"correct\": %llu, "
           "\"other\": %llu}\n",
           SGPR_SOFF ? "sgpr" : "zero", W, (long long)waves * iters, (long long)pieces, h[0], h[1], h[2], h[3]);
}

int main(int argc, char **argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 8;
    int iters = argc > 2 ? atoi(argv[2]) : 256;
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int waves = cus * 4 * wps;
    while ((int64_t)waves * iters * 1024 >= ((int64_t)1 << 32) - (1 << 20)) iters /= 2; // slot offsets stay below 4 GB (one buffer resource)
    uint32_t *buf;
    unsigned long long *d_out;
    CHECK(hipMalloc((void **)&buf, (size_t)waves * iters * 1024));
    CHECK(hipMalloc((void **)&d_out, 32));
    fprintf(stderr, "%d compute units, %d waves per SIMD, %d waves, %d stores each\n", cus, wps, waves, iters);
    run<0, true>(buf, waves, iters, d_out);
    run<1, true>(buf, waves, iters, d_out);
    run<2, true>(buf, waves, iters, d_out);
    run<3, true>(buf, waves, iters, d_out);
    run<0, false>(buf, waves, iters, d_out);
    run<1, false>(buf, waves, iters, d_out);
    run<2, false>(buf, waves, iters, d_out);
    return 0;
}

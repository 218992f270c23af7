// sgd_device.hpp -- device-side helpers shared by the SGD kernel translation units (mf_sgd_kernels.hip,
// chain_kernels.hip): DPP row reductions and the per-model container traits.  Internal header.
#pragma once
#include <hip/hip_runtime.h>

#include "mf_sgd_kernels.hpp"

namespace cmi {

// ---------------------------------------------------------------------------------------------
// cross-lane helpers
// ---------------------------------------------------------------------------------------------

// DPP row_ror:n -- lane l of each 16-lane row reads lane (l - n) mod 16 of the same row.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}

// Sum over the 16 lanes of a DPP row; every lane ends with the bit-identical total (the rotation
// tree pairs the same operands in every lane, and fp add is commutative).
__device__ __forceinline__ float row_sum16(float x) {
    x += dpp_f32<0x128>(x); // row_ror:8
    x += dpp_f32<0x124>(x); // row_ror:4
    x += dpp_f32<0x122>(x); // row_ror:2
    x += dpp_f32<0x121>(x); // row_ror:1
    return x;
}

template <typename T>
__device__ __forceinline__ T wave_sum64(T x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}

template <int MODEL>
struct Traits {
    static constexpr bool has_bu = MODEL == BIASEDMF || MODEL == CAMF_C || MODEL == CAMF_CI;
    static constexpr bool has_bj = MODEL == BIASEDMF || MODEL == CAMF_C || MODEL == CAMF_CU;
    static constexpr bool has_bc = MODEL == CAMF_C;
    static constexpr bool has_ic = MODEL == CAMF_CI || MODEL == CAMF_CUCI;
    static constexpr bool has_uc = MODEL == CAMF_CU || MODEL == CAMF_CUCI;
    static constexpr bool has_ctx = !(MODEL == BIASEDMF || MODEL == PMF); // iterates the contextual matrix
};


// Sum over the LPT (4, 8 or 16) lanes of a group that shares one tuple in the small-k kernels: quad permutes, then row
// (half-)mirrors; every lane of the group ends with the same total.
template <int LPT>
__device__ __forceinline__ float group_sum(float x) {
    x += dpp_f32<0xB1>(x); // quad_perm [1,0,3,2]
    x += dpp_f32<0x4E>(x); // quad_perm [2,3,0,1]
    if (LPT >= 8) x += dpp_f32<0x141>(x);  // row_half_mirror: lane i <- lane 7-i of its half row (the other quad's total)
    if (LPT >= 16) x += dpp_f32<0x140>(x); // row_mirror: lane i <- lane 15-i (the other half's total)
    return x;
}

// fp64 flavour of the 16-lane DPP row sum (both dwords rotated with the same control)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum16(double x) {
    x += dpp_f64<0x128>(x);
    x += dpp_f64<0x124>(x);
    x += dpp_f64<0x122>(x);
    x += dpp_f64<0x121>(x);
    return x;
}

} // namespace cmi

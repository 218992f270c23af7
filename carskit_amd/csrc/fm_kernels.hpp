// fm_kernels.hpp -- launch interface of fm_kernels.hip (internal)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmi {

struct FmArgs {
    // model (fp64, the reference's precision)
    double *w0;   // 1
    double *w;    // p
    double *V;    // p x k row-major
    // per-rating state, STORAGE order = ratings sorted by user (stable), so user supports are contiguous
    double *err;  // n
    double *Qt;   // k x n  (column f of the reference's Q is contiguous)
    const int32_t *u, *j, *ctx;
    const double *r;
    // supports: field 0 users (sup[0] == nullptr: contiguous), 1 items, 2 context features
    const int32_t *sup[3];
    const int64_t *sup_off[3];
    int32_t field_count[3];
    double *part; // [num | den] of the phase's field, or w0 scratch
    int64_t n, global_size;
    int32_t k, n_users, n_items, n_conds;
    double xc;    // 1 / numContextDims
    double regLw, regLf;
};

hipError_t fm_launch_field(const FmArgs &a, int field, int f, int mode /*0 reduce, 1 apply, 2 fused*/, hipStream_t s);
hipError_t fm_launch_w0_reduce(const FmArgs &a, double *scratch, hipStream_t s);
hipError_t fm_launch_w0_apply(const FmArgs &a, hipStream_t s);
hipError_t fm_launch_init(const FmArgs &a, hipStream_t s);
hipError_t fm_launch_predict(const FmArgs &a, int64_t n, const int32_t *tu, const int32_t *tj, const int32_t *tc,
                             int bound, double lo, double hi, double *out, hipStream_t s);

} // namespace cmi

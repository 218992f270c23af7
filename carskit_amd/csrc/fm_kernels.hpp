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
    // per-coordinate working table (p entries, L2/MALL-resident gathers): .x = column f of V (the factor being swept),
    // .y = the coordinate delta of the last item / context phase that is not yet folded into the errors
    double2 *tab;
    // per-rating state, STORAGE order = ratings sorted by user (stable), so user supports are contiguous:
    // .x = errors[i], lazily maintained (true error = .x + tab[item].y + xc * tab[ctx].y while `pending` says so);
    // .y = V[user of i][f] as left by the user phase of factor f (what the item / context phases need of the user)
    double2 *R;
    const int32_t *u, *j, *ctx;
    const double *r;
    // supports: field 0 users (contiguous storage ranges), 1 items, 2 context features: storage positions + the
    // other two feature ids of each supporting rating (so a reduce pass gathers only err[])
    const int32_t *sup[3], *sup_a[3], *sup_b[3];
    const int64_t *sup_off[3];
    int32_t field_count[3];
    double *part; // [num | den] of the phase's field, or w0 scratch
    int64_t n, global_size;
    int32_t k, n_users, n_items, n_conds;
    int32_t pending; // pend_j / pend_c hold non-zero deltas
    double xc;       // 1 / numContextDims
    double regLw, regLf;
};

// f < 0: linear weights w; f >= 0: column f of V (a.col must hold it).  mode 0 reduce -> part, 1 apply <- part, 2 fused.
hipError_t fm_launch_field(const FmArgs &a, int field, int f, int mode, hipStream_t s);
hipError_t fm_launch_col_load(const FmArgs &a, int f, hipStream_t s); // tab[l].x = V[l][f]
hipError_t fm_launch_w0_reduce(const FmArgs &a, double *scratch, hipStream_t s);
hipError_t fm_launch_w0_apply(const FmArgs &a, hipStream_t s); // also folds the pending deltas into err
hipError_t fm_launch_flush(const FmArgs &a, hipStream_t s);    // err += pending deltas
hipError_t fm_launch_uval(const FmArgs &a, hipStream_t s);     // R[i].y = tab[user of i].x
hipError_t fm_launch_init(const FmArgs &a, hipStream_t s);
hipError_t fm_launch_predict(const FmArgs &a, int64_t n, const int32_t *tu, const int32_t *tj, const int32_t *tc,
                             int bound, double lo, double hi, double *out, hipStream_t s);

} // namespace cmi

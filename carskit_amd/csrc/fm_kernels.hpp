// fm_kernels.hpp -- launch interface of fm_kernels.hip (internal)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmi {

// One rating as a phase streams it: the error as computed by cmi_fm_init (never rewritten by a sweep) and the rating's
// OTHER two feature ids.  a = item (user order) / user (item order) / user (context order); c = context-combination id
// (user and item order) / item (context order).
struct FmRec {
    double err0;
    int32_t a, c;
};

// How a chunk of the stream is reduced (one wave per chunk).
//   stream chunk: pieces [piece0, piece0 + n) (n <= 64, every piece <= FM_SHORT records), records [rec0, rec1) (<= 256):
//                 a lane per piece sums its records left to right out of LDS -> partial[piece]
//   vector chunk: n < 0: records [rec0, rec1) of the ONE long piece piece0, all lanes -> partial[grid + (-n - 1)]
struct FmChunk {
    int32_t piece0, n, rec0, rec1;
};

constexpr int FM_SHORT = 64;      // longest piece a single lane sums
constexpr int FM_CHUNK = 512;     // records per stream chunk
constexpr int FM_VECTOR = 2048;   // records per vector chunk

// The ratings in the order one FIELD streams them: sorted by (slice of the other id, this field's coordinate), so a
// coordinate's support is S contiguous pieces and the table entries a slice gathers stay L2-resident.
// piece p = slice * count + coordinate; piece_off[p] .. piece_off[p + 1] its records.
struct FmOrder {
    const FmRec *rec;
    const int32_t *piece_off; // S * count + 1
    const FmChunk *chunks;
    const int32_t *xoff; // count + 1: coordinate l's extra partial slots are grid + [xoff[l], xoff[l + 1])  (vector chunks)
    double2 *partial;    // S * count + n_x
    int32_t n_chunks, count, S, n_x;
    int64_t n_rec;
};

struct FmArgs {
    // model (fp64, the reference's precision)
    double *w0;   // 1
    double *d0;   // 1: sum of the w0 deltas since cmi_fm_init
    double *w;    // p
    double *V;    // p x k row-major (the API's layout; cmi_fm_init, predict and the rankings read it)
    double *Vt;   // k x p: the sweeps' working copy -- a column of V is contiguous here (fm_col_load, fm_finish_kernel)
    // per-coordinate working table: .x = column f of V (the factor being swept), .y = D[l] = sum of the coordinate's
    // deltas (over w and every column of V) since cmi_fm_init.  The reference's errors[i] (FM.java:133-136,165,188,208) is
    //   err0[i] + d0 + D[user] + D[item] + xc * D[ctx feature]
    // so no phase ever writes per-rating data.
    double2 *tab;
    FmOrder ord[3]; // field 0 users, 1 items, 2 context features
    double *part;   // [num | den] of the phase's field, or w0 scratch
    // the ratings in user order as plain arrays (init only)
    const int32_t *u, *j, *ctx;
    const double *r;
    const int32_t *i2u, *c2u; // item-order / context-order record -> user-order record (init only)
    int64_t n, global_size;
    int32_t k, n_users, n_items, n_conds;
    double xc; // 1 / numContextDims
    double regLw, regLf;
};

// f < 0: linear weights w; f >= 0: column f of V (a.tab[].x must hold it).
hipError_t fm_launch_reduce(const FmArgs &a, int field, int f, hipStream_t s);              // -> ord[field].partial
hipError_t fm_launch_finish(const FmArgs &a, int field, int f, int mode, hipStream_t s);    // 0 partial -> part; 1 part -> update; 2 both
hipError_t fm_launch_col_load(const FmArgs &a, int f, hipStream_t s);                       // tab[l].x = Vt[f][l]
hipError_t fm_launch_transpose(const double *src, double *dst, int64_t rows, int64_t cols, hipStream_t s); // dst[c][r] = src[r][c]
hipError_t fm_launch_w0_reduce(const FmArgs &a, double *scratch, hipStream_t s);            // part[0] = sum(err_i - w0)
hipError_t fm_launch_w0_apply(const FmArgs &a, hipStream_t s);
hipError_t fm_launch_init(const FmArgs &a, hipStream_t s);
hipError_t fm_launch_predict(const FmArgs &a, int64_t n, const int32_t *tu, const int32_t *tj, const int32_t *tc,
                             int bound, double lo, double hi, double *out, hipStream_t s);

} // namespace cmi

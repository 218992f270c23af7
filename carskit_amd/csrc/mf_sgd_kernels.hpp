// mf_sgd_kernels.hpp -- launch interface between the C-ABI host code (cmi_api.cpp) and the gfx950
// kernels (mf_sgd_kernels.hip).  Internal header; the public surface is include/carskit_mi355x.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmi {

enum Model { BIASEDMF = 0, CAMF_C = 1, CAMF_CI = 2, CAMF_CU = 3, CAMF_CUCI = 4, PMF = 5,
             SVDPP = 6, CAMF_ICS = 7, CAMF_LCS = 8, CAMF_MCS = 9 }; // 6..9: ext_kernels.hip (serial only)

// Device-resident hyper-parameters, rewritten before every epoch by set_hparams (so a captured
// hipGraph of level launches can be replayed with a new learning rate).
struct HParams {
    double lr, regU, regI, regB, regC, gm;
};

// Everything a training kernel needs; T = element type of the model state in HBM.
template <typename T>
struct SgdArgs {
    T *P, *Q;                  // [n_users x k], [n_items x k] row-major, row = k contiguous elements
    T *userBias, *itemBias;    // [n_users], [n_items]
    T *condBias;               // [n_conds]
    T *ucBias, *icBias;        // [n_users x n_conds], [n_items x n_conds]
    const int32_t *su, *sj;    // tuple stream in SCHEDULE order (levels concatenated)
    const T *sr;               // ratings, same order
    const int32_t *sconds;     // [n x dmax] condition ids of each tuple, -1 padded
    const HParams *hp;
    double *loss_part;         // one slot per workgroup of the epoch (deterministic reduction)
    int32_t k, n_conds, dmax;
    // hub-chain levels, large spoke table (round 3): the spoke row of the tuple at stream position p lives in arena[p] -- read
    // sequentially in schedule order -- and goes back to arena[next_pos[p]], the slot of the same row's next tuple (chain_kernels.hip).
    // null: rows are read from / written to the model table itself
    T *arena = nullptr;
    const int32_t *next_pos = nullptr;
    // owner epoch: the tag a spoke record carries when the epoch begins (a row's tag = tag0 + its updates so far this epoch).  A new
    // value every epoch, so that a copy of a record left anywhere from an EARLIER epoch can never pass for the one a tuple waits for.
    uint32_t owner_tag0 = 0;
#ifdef CMI_OWNER_TRACE
    double *trace = nullptr; // debug builds (make TRACE=1): per-tuple inputs and outputs of the owner epoch, owner_kernels.hip
#endif
};

// spoke arena <-> model table (n_rows rows of k elements; first_pos[row] = stream position of the row's first tuple, -1: none)
template <typename T>
hipError_t launch_arena_scatter(const T *table, T *arena, const int32_t *first_pos, int64_t n_rows, int k, hipStream_t s);
template <typename T>
hipError_t launch_arena_gather(T *table, const T *arena, const int32_t *first_pos, int64_t n_rows, int k, hipStream_t s);

// CAMF_C as one software-pipelined wave (camfc_pipe.hip): rows of tuple t + D requested while tuple t is computed
bool camfc_pipe_supported(int k, int n_conds, int dmax);
template <typename T>
hipError_t launch_camfc_pipe(const SgdArgs<T> &a, int64_t n, double *loss_out, hipStream_t s);

struct LaunchCfg {
    int model;
    bool strict; // left-to-right dot (DenseMatrix.rowMult order) + reference loss order
};

// number of workgroups a level of `count` tuples occupies (= loss_part slots it writes)
int level_blocks_f32_fast(int k, int count);
int level_blocks_generic(int count);
bool has_fast_path(int k, int dmax, bool f64, const LaunchCfg &cfg);
// all narrow tail levels of a schedule in one launch (one workgroup, a barrier per level); tail_off = n_tail+1 device offsets
template <typename T>
hipError_t launch_tail(const SgdArgs<T> &a, const LaunchCfg &cfg, const int64_t *tail_off, int n_tail, int64_t slot,
                       hipStream_t s);
// fp32 state: kind 0 = generic arithmetic, 1 = the float4 level kernels', 2 = the small-k level kernels' (same bits as the
// level launches the run replaces)
hipError_t launch_tail_f32(const SgdArgs<float> &a, const LaunchCfg &cfg, int kind, const int64_t *tail_off, int n_tail,
                           int64_t slot, hipStream_t s);
// CAMF_C over conflict-free CRS blocks (<= 64 tuples sharing no user and no item; blk_off = n_blocks+1 device offsets):
// exact, with the per-tuple gather/dot/update parallel inside a block and only the scalar condBias chain sequential
size_t camfc_blocks_lds(int n_conds, int dmax, size_t esize);
template <typename T>
hipError_t launch_camfc_blocks(const SgdArgs<T> &a, const int32_t *blk_off, int n_blocks, double *loss_out, hipStream_t s);
// small-k fast path (fp32 state, k < 64): 4 / 8 / 16 lanes per tuple
bool has_small_path(int k, int dmax, bool f64, const LaunchCfg &cfg);
int level_blocks_small(int k, int dmax, int count);
hipError_t launch_level_small_f32(const SgdArgs<float> &a, const LaunchCfg &cfg, int64_t begin, int count, int64_t slot0,
                                  hipStream_t s);

// one dependency level: tuples [begin, begin+count) of the schedule run concurrently
hipError_t launch_level_fast_f32(const SgdArgs<float> &a, const LaunchCfg &cfg, int64_t begin, int count,
                                 int64_t slot0, hipStream_t s);
template <typename T>
hipError_t launch_level_generic(const SgdArgs<T> &a, const LaunchCfg &cfg, int64_t begin, int count, int64_t slot0,
                                hipStream_t s);
// Explicit hipGraph nodes (the two-lane schedule needs a DAG, not a linear capture): one fast-path level segment,
// and the two-stage loss reduction.  count == 0 adds an empty node so dependency chains stay uniform.
// Hub-chain level kernel (chain_kernels.hip): one 16-lane group walks a unit of the chain schedule with the hub row on chip.
// units [ubegin, ubegin+count) of unit_off (n_units+1 device offsets into the tuple stream) form one level.
bool has_chain_path(int model, int k, int dmax, int n_conds, bool f64, bool strict);
size_t chain_lds_bytes(int model, int n_conds, int dmax, bool f64, bool hub_is_item);
int chain_groups_per_block(int k, int dmax, bool f64);   // units per 256-thread workgroup (16; 16 / 32 / 64 for fp32 k < 64)
int chain_level_blocks(int k, int dmax, bool f64, int count);
// a run of n_levels narrow chain levels (<= 64 units each) walked by ONE workgroup; lvl_off = device offsets (unit indices) of the
// run's levels, n_levels + 1 entries
template <typename T>
hipError_t launch_chain_level(const SgdArgs<T> &a, const LaunchCfg &cfg, bool hub_is_item, const int32_t *unit_off, int64_t ubegin,
                              int count, int64_t slot0, hipStream_t s);

// one wavefront walks tuples [0, n) in order (the reference's sequential semantics); loss -> loss_out[0] (already *0.5)
template <typename T>
hipError_t launch_serial(const SgdArgs<T> &a, const LaunchCfg &cfg, int64_t n, double *loss_out, hipStream_t s);

hipError_t launch_set_hparams(HParams *dst, HParams v, hipStream_t s);
// loss_out[0] = 0.5 * sum(loss_part[0..n_slots)) in a fixed order; scratch holds >= 256 doubles
hipError_t launch_reduce_loss(const double *loss_part, int64_t n_slots, double *scratch, double *loss_out,
                              hipStream_t s);

// predict / evalRatings
template <typename T>
struct EvalArgs {
    const T *P, *Q, *userBias, *itemBias, *condBias, *ucBias, *icBias;
    const int32_t *u, *j, *ctx;       // n tuples (ctx may be null for BiasedMF)
    const double *r;                  // may be null (predict only)
    const int32_t *ctx_ptr, *ctx_conds;
    double *preds;                    // may be null
    double *part;                     // [blocks x 5] partial sums (abs, sq, rabs, rsq, count); may be null
    double gm, lo, hi, min_rate;
    int32_t k, n_conds, bound, model;
};
int eval_blocks(int64_t n);
template <typename T>
hipError_t launch_eval(const EvalArgs<T> &a, int64_t n, hipStream_t s);

// multi-GPU exchange passes (element type per f64): bucket = state - snap ; state = snap = snap + scale * bucket
hipError_t launch_delta_pack(const void *state, const void *snap, void *bucket, int64_t n, bool f64, hipStream_t s);
hipError_t launch_delta_apply(void *state, void *snap, const void *bucket, double scale, int64_t n, bool f64, hipStream_t s);

// ---- SVD++ / CAMF_ICS / CAMF_LCS / CAMF_MCS (ext_kernels.hip): serial only ----
template <typename T>
struct ExtArgs {
    T *P, *Q, *userBias, *itemBias;  // userBias / itemBias / Y: SVD++
    T *Y;                            // [n_items x k]
    T *cc;                           // CAMF_ICS: [n_conds x n_conds], kept symmetric
    T *cf;                           // CAMF_LCS: [n_conds x num_f]
    T *cv;                           // CAMF_MCS: [n_conds]
    const int32_t *su, *sj;          // tuples in the reference's (CRS) order
    const T *sr;
    const int32_t *sconds;           // [n x dmax], -1 padded
    const int32_t *empty_conds;      // EmptyContextConditions: the i-th condition of a context pairs with empty_conds[i]
    const int32_t *ui_ptr, *ui_items; // SVD++: items of every user in the 2-D train matrix (userItemsCache), ascending
    const HParams *hp;
    double upbound, lowbound;        // CAMF_MCS.java:47-48
    int32_t k, n_conds, dmax, num_f, n_empty;
};
template <typename T>
hipError_t launch_ext_serial(const ExtArgs<T> &a, int model, bool strict, int64_t n, double *loss_out, hipStream_t s);
// SVD++ with a workgroup per chain link, the user's rows resident in LDS (svdpp_team.hip)
bool svdpp_team_supported(int k);
template <typename T>
hipError_t launch_svdpp_team(const ExtArgs<T> &a, int64_t n, double *loss_out, hipStream_t s);

template <typename T>
struct ExtEvalArgs {
    const T *P, *Q, *userBias, *itemBias, *Y, *cc, *cf, *cv;
    const int32_t *u, *j, *ctx;
    const double *r;
    const int32_t *ctx_ptr, *ctx_conds, *empty_conds, *ui_ptr, *ui_items;
    double *preds, *part;
    double gm, lo, hi, min_rate;
    int32_t k, n_conds, num_f, n_empty, bound, model;
};
template <typename T>
hipError_t launch_ext_eval(const ExtEvalArgs<T> &a, int64_t n, hipStream_t s);
// operands of the ranking evaluation for these models (see ext_kernels.hip)
template <typename T>
hipError_t launch_ext_rank_items(const ExtEvalArgs<T> &a, const int32_t *cand, int nc, T *B, int kp, hipStream_t s);
template <typename T>
hipError_t launch_ext_rank_queries(const ExtEvalArgs<T> &a, const int32_t *qu, const int32_t *qc, int nq, T *A, T *row_const, int kp,
                                   hipStream_t s);

// dtype conversion for cmi_set_state / cmi_get_state staging
hipError_t launch_convert(const void *src, int src_f64, void *dst, int dst_f64, int64_t n, hipStream_t s);


// ---- owner (dataflow) epoch for heavy-tailed degrees (owner_kernels.hip; schedule: build_owner_schedule) ----
template <int NCW>
struct OwnerRecT {      // one tuple of an owner's list, read through scalar loads (24 + 8 NCW bytes)
    uint32_t off;       // byte offset of the spoke row's tagged record in the record table (< 4 GB: one buffer resource)
    int32_t hub;        // the row the owner keeps
    uint32_t want;      // tag the spoke record must carry (= its update count before this tuple)
    uint32_t flags;     // OWN_* (level_schedule.hpp)
    union {
        double d;       // fp64 state
        float f;        // fp32 state
    } rating;
    uint64_t mask[NCW]; // bit c of word w = condition 64 w + c is in the tuple's context (each word is used as a lane mask)
};
inline size_t owner_rec_bytes(int ncw) { return 24 + 8 * (size_t)ncw; }
int owner_mask_words(int model, int n_conds);                                             // NCW for a model: 1, 2 or 6
bool has_owner_path(int model, int k, int n_conds, bool f64, bool strict);
int owner_depth();                                                                        // read-ahead distance of the kernel
int64_t owner_record_stride(int model, int k, int n_conds, bool f64, bool hub_is_item);   // granules (8 bytes) per spoke record
int owner_grid_waves(int device, int model, int n_conds, int k, bool f64, bool hub_is_item);           // owners that are resident together
// tag pass + the persistent epoch + untag pass; loss partials in a.loss_part[0 .. n_owners)
template <typename T>
// owners [0, n_team) run as teams of three wavefronts (one workgroup each), the rest four to a workgroup
hipError_t launch_owner_epoch(const SgdArgs<T> &a, int model, bool hub_is_item, bool strict, const void *recs, const int64_t *own_off,
                              int n_owners, int n_team, void *tagged, int64_t stride, int n_spokes, int *error, uint32_t tag0, hipStream_t s);

} // namespace cmi

// cmi_instance.hpp -- the object behind a cmi_handle, shared by the translation units of libcarskit_mi355x.so.
#pragma once
#include "rank_host.hpp"
#include "../../include/carskit_mi355x.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "mf_sgd_kernels.hpp"

struct cmi_instance {
    int model = 0, k = 0, n_users = 0, n_items = 0, n_conds = 0, device = 0;
    unsigned flags = 0;
    bool f64 = false, serial = false, strict = false, use_graph = true, fast = false, small = false;
    bool chain = false, chain_hub_item = true; // hub-chain level schedule: level_off holds UNIT indices, d_unit_off the units
    int32_t *d_unit_off = nullptr;
    int64_t n_units = 0;
    // spoke arena of the hub-chain schedule (SgdArgs::arena): one slot per tuple, in schedule order.  While arena_valid the live value of
    // every spoke row with tuples sits in the slot of its FIRST tuple; the model table (state[arena_which]) is only current while
    // table_valid -- every reader of the table goes through cmi_sync_table_from_arena first
    bool arena_on = false, arena_valid = false, table_valid = true;
    bool arena_probe_table = false;  // test hook: the probe's verdict is forced to "table"
    bool arena_probe = false;        // the first training call still has to choose between table and arena (cmi_api.cpp arena_probe)
    int arena_which = 0;             // CMI_STATE_P (hub = item) or CMI_STATE_Q (hub = user)
    void *d_arena = nullptr;
    int32_t *d_next = nullptr, *d_first = nullptr;
    // owner (dataflow) schedule: one persistent launch, d_own_recs = the owners' lists, d_tagged = the spoke side's tagged records
    bool want_owner = false, owner = false, owner_hub_item = true;
    uint32_t owner_epoch_seq = 0; // owner epochs launched so far (the epoch's tag base derives from it)
    bool owner_busy = false; // the last owner epoch was not launched: the device's owner-epoch lock could not be taken (CMI_E_BUSY)
    const char *owner_busy_why = "";
    int device_share = 1; // cmi_set_device_share: instances training concurrently on this device (sizes the persistent kernels' grids)
    int n_owners = 0, n_team = 0; // owners [0, n_team) run as teams of three wavefronts
    bool owner_stalled = false;   // an owner epoch hit its wait bound: the model state is invalid (sticky until cmi_set_ratings)
    void *d_own_recs = nullptr; // cmi::OwnerRecT<NCW>[]
    int64_t *d_own_off = nullptr;
    void *d_tagged = nullptr;
    int64_t own_stride = 0;
#ifdef CMI_OWNER_TRACE
    double *d_trace = nullptr; // debug builds (make TRACE=1): cmi_debug_owner_trace / cmi_debug_owner_trace_dump
    size_t trace_doubles = 0;
#endif
    std::string err;
    std::string sched_note; // why a slower schedule than the data calls for is running (cmi_schedule_note); empty = nothing to say
    hipStream_t stream = nullptr;
    void *state[CMI_STATE_COUNT] = {};
    int64_t state_count[CMI_STATE_COUNT] = {};
    // tuple stream (schedule order)
    int64_t n = 0;
    int dmax = 0;
    int32_t n_ctx = 0;
    int32_t *d_su = nullptr, *d_sj = nullptr, *d_sconds = nullptr, *d_ctx_ptr = nullptr, *d_ctx_conds = nullptr;
    void *d_sr = nullptr;
    int32_t *d_flow_err = nullptr; // owner epoch: stall flag + counters
    int64_t ctx_nnz = 0;
    std::vector<int64_t> level_off, slot_off;
    int64_t n_launches = 0, n_tail = 0; // launches per epoch; levels that live inside narrow runs
    std::vector<int32_t> tail_len;      // per level: >0 = a narrow run of that many levels starts here (one launch), -1 = inside one
    int64_t *d_tail_off = nullptr;
    std::vector<int32_t> blk_off; // CAMF_C: conflict-free CRS blocks (empty: the serial wave is used)
    int32_t *d_blk_off = nullptr;
    int64_t n_slots = 0, max_level = 0, tuple_bytes = 0, sched_levels = 0;
    double *d_loss_part = nullptr, *d_scratch = nullptr, *d_loss = nullptr;
    cmi::HParams *d_hp = nullptr;
    cmi::HParams hp{0, 0, 0, 0, 0, 0};
    double *h_loss = nullptr; // pinned
    hipGraphExec_t graph_exec = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool have_ratings = false, epoch_timed = false;
    double last_loss = 0.0;
    // resident test tuples (cmi_set_eval_ratings)
    int32_t *d_eu = nullptr, *d_ej = nullptr, *d_ectx = nullptr;
    double *d_er = nullptr, *d_epart = nullptr;
    int64_t n_eval = 0;
    // multi-GPU exchange of the item-side containers (cmi_exchange_*): bucket and snapshot share one layout
    void *d_xbucket = nullptr, *d_xsnap = nullptr;
    int64_t x_count = 0;
    std::vector<int> x_which;       // containers in the bucket, in order
    std::vector<int64_t> x_off;     // their element offsets (16-byte aligned segments)
    // SVD++ / CAMF_*CS (ext_kernels.hip)
    int num_f = 0, n_ctx_dims = 1;
    bool sim_params_set = false; // cmi_set_sim_params has run (an EMPTY EmptyContextConditions list is a valid setting)
    std::vector<int32_t> empty_conds;
    int32_t *d_empty = nullptr, *d_ui_ptr = nullptr, *d_ui_items = nullptr;
    void *comm = nullptr;        // ncclComm_t of cmi_comm_init (one-process-per-GPU jobs); group_api.cpp owns the type
    int comm_rank = 0, comm_world = 0;
    hipEvent_t evx0 = nullptr, evx1 = nullptr; // around the most recent cmi_comm_exchange (cmi_comm_last_exchange_ms)
    bool exchange_timed = false;
    cmi::RankWorkspace rank_ws;  // cmi_eval_rankings' device / pinned buffers, reused by the next evaluation
    float last_rank_ms = 0.f;    // device time of the most recent cmi_eval_rankings scoring loop (HIP events)
    double last_rank_flops = 0.0; // 2 * queries * candidates * padded operand length of that loop
};

#define CMI_FAIL(h, code, ...)                                                                          \
    do {                                                                                                \
        char buf_[512];                                                                                 \
        snprintf(buf_, sizeof buf_, __VA_ARGS__);                                                       \
        (h)->err = buf_;                                                                                \
        return (code);                                                                                  \
    } while (0)

#define CMI_HIP(h, expr)                                                                                \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) CMI_FAIL(h, CMI_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

bool cmi_model_has(int model, int which); // which containers a model owns (cmi_api.cpp)
// evaluation-side view of an SVD++ / CAMF_*CS instance (cmi_api.cpp); the tuple / output pointers may be null for the ranking operands
template <typename T>
cmi::ExtEvalArgs<T> cmi_ext_eval_args(cmi_instance *h, const int32_t *du, const int32_t *dj, const int32_t *dctx, const double *dr,
                                      double *dpreds, double *dpart, int bound, double lo, double hi, double min_rate);

// IterativeRecommender.isConverged + updateLRate around an epoch function (cmi_api.cpp; shared by cmi_train_from and cmi_group_train_from)
int cmi_train_loop(const std::function<int(double, double *)> &epoch, std::string &err, int first_iter, double prev_loss, int num_iters,
                   double init_lrate, double max_lrate, int bold_driver, double decay, int early_stop, double *losses, double *lrates,
                   int *iters_run, double *final_lrate);
int cmi_eval_sums(cmi_instance *h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, const double *r, double min_rate,
                  double max_rate, double sums[5]);

// the sums of cmi_eval_resident before they are turned into measures (a group adds them over its shards)
int cmi_eval_resident_sums(cmi_instance *h, double min_rate, double max_rate, double sums[5]);
// cmi_comm_* (group_api.cpp, which owns the RCCL types): destroy the handle's communicator, if any
void cmi_comm_release(cmi_instance *h);

// spoke arena (cmi_api.cpp): make the model table current (gather from the arena) / say that the table was rewritten (arena stale)
int cmi_sync_table_from_arena(cmi_instance *h);

// cmi_api.cpp -- implementation of include/carskit_mi355x.h: instance handle, device memory,
// schedule construction, hipGraph capture of the per-level launches, the isConverged/updateLRate
// loop, predict/eval.  Compiled by hipcc together with mf_sgd_kernels.hip into libcarskit_mi355x.so.
// There is deliberately no CPU code path for any compute entry point.
#include "../../include/carskit_mi355x.h"
#include "env_knobs.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "cmi_instance.hpp"
#include "host_pool.hpp"
#include "level_schedule.hpp"
#include "sched_device.hpp"
#include "mf_sgd_kernels.hpp"

using namespace cmi;

static thread_local std::string g_create_err;

#include "owner_gate.hpp"

bool cmi_model_has(int model, int which) {
    switch (which) {
    case CMI_STATE_P:
    case CMI_STATE_Q: return true;
    case CMI_STATE_USER_BIAS: return model == CMI_MODEL_BIASEDMF || model == CMI_MODEL_CAMF_C || model == CMI_MODEL_CAMF_CI || model == CMI_MODEL_SVDPP;
    case CMI_STATE_ITEM_BIAS: return model == CMI_MODEL_BIASEDMF || model == CMI_MODEL_CAMF_C || model == CMI_MODEL_CAMF_CU || model == CMI_MODEL_SVDPP;
    case CMI_STATE_Y: return model == CMI_MODEL_SVDPP;
    case CMI_STATE_CC_MATRIX: return model == CMI_MODEL_CAMF_ICS;
    case CMI_STATE_CF_MATRIX: return model == CMI_MODEL_CAMF_LCS;
    case CMI_STATE_C_VECTOR: return model == CMI_MODEL_CAMF_MCS;
    case CMI_STATE_COND_BIAS: return model == CMI_MODEL_CAMF_C;
    case CMI_STATE_UC_BIAS: return model == CMI_MODEL_CAMF_CU || model == CMI_MODEL_CAMF_CUCI;
    case CMI_STATE_IC_BIAS: return model == CMI_MODEL_CAMF_CI || model == CMI_MODEL_CAMF_CUCI;
    }
    return false;
}

static int64_t state_elems(const cmi_instance *h, int which) {
    switch (which) {
    case CMI_STATE_P: return (int64_t)h->n_users * h->k;
    case CMI_STATE_Q: return (int64_t)h->n_items * h->k;
    case CMI_STATE_USER_BIAS: return h->n_users;
    case CMI_STATE_ITEM_BIAS: return h->n_items;
    case CMI_STATE_COND_BIAS: return h->n_conds;
    case CMI_STATE_UC_BIAS: return (int64_t)h->n_users * h->n_conds;
    case CMI_STATE_IC_BIAS: return (int64_t)h->n_items * h->n_conds;
    case CMI_STATE_Y: return (int64_t)h->n_items * h->k;
    case CMI_STATE_CC_MATRIX: return (int64_t)h->n_conds * h->n_conds;
    case CMI_STATE_CF_MATRIX: return (int64_t)h->n_conds * h->num_f; // 0 until cmi_set_sim_params
    case CMI_STATE_C_VECTOR: return h->n_conds;
    }
    return 0;
}

static size_t esize(const cmi_instance *h) { return h->f64 ? 8 : 4; }
static bool is_ext_model(int model) { return model >= CMI_MODEL_SVDPP && model <= CMI_MODEL_CAMF_MCS; }
static bool is_2d_model(int model) { return model == CMI_MODEL_BIASEDMF || model == CMI_MODEL_PMF || model == CMI_MODEL_SVDPP; }

extern "C" int cmi_abi_version(void) { return CMI_ABI_VERSION; }

extern "C" int cmi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *cmi_last_error(cmi_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

static void free_eval_set(cmi_instance *h) {
    void *ptrs[] = {h->d_eu, h->d_ej, h->d_ectx, h->d_er, h->d_epart};
    for (void *p : ptrs)
        if (p) hipFree(p);
    h->d_eu = h->d_ej = h->d_ectx = nullptr;
    h->d_er = h->d_epart = nullptr;
    h->n_eval = 0;
}

// the resident test tuples index the context table of the ratings they were uploaded against: they go with it
static void free_ratings(cmi_instance *h) {
    if (h->arena_on && h->arena_valid && !h->table_valid) cmi_sync_table_from_arena(h); // the live rows are in the arena: bring them home first
    for (void **p : {(void **)&h->d_arena, (void **)&h->d_next, (void **)&h->d_first})
        if (*p) {
            hipFree(*p);
            *p = nullptr;
        }
    h->arena_on = h->arena_valid = false;
    h->arena_probe = false;
    h->table_valid = true;
    free_eval_set(h);
    if (h->graph_exec) {
        hipGraphExecDestroy(h->graph_exec);
        h->graph_exec = nullptr;
    }
    void *ptrs[] = {h->d_su, h->d_sj, h->d_sconds, h->d_ctx_ptr, h->d_ctx_conds, h->d_sr, h->d_loss_part,
                    h->d_flow_err, h->d_tail_off, h->d_blk_off, h->d_unit_off,
                    h->d_ui_ptr, h->d_ui_items, h->d_own_recs, h->d_own_off, h->d_tagged};
    h->d_own_recs = nullptr;
    h->d_own_off = nullptr;
    h->d_tagged = nullptr;
    h->owner = false;
    h->owner_stalled = false;
    h->d_ui_ptr = h->d_ui_items = nullptr;
    for (void *p : ptrs)
        if (p) hipFree(p);
    h->d_su = h->d_sj = h->d_sconds = h->d_ctx_ptr = h->d_ctx_conds = nullptr;
    h->d_flow_err = nullptr;
    h->d_tail_off = nullptr;
    h->d_blk_off = nullptr;
    h->d_unit_off = nullptr;
    h->chain = false;
    h->n_units = 0;
    h->blk_off.clear();
    h->n_launches = h->n_tail = 0;
    h->tail_len.clear();
    h->d_sr = nullptr;
    h->d_loss_part = nullptr;
    h->have_ratings = false;
    h->n = 0;
}

extern "C" int cmi_destroy(cmi_handle h) {
    if (!h) return CMI_OK;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    free_ratings(h);
    cmi_comm_release(h);
    h->rank_ws.release();
    for (void *&p : h->state)
        if (p) {
            hipFree(p);
            p = nullptr;
        }
    if (h->d_empty) hipFree(h->d_empty);
    if (h->d_xbucket) hipFree(h->d_xbucket);
    if (h->d_xsnap) hipFree(h->d_xsnap);
    if (h->d_scratch) hipFree(h->d_scratch);
    if (h->d_loss) hipFree(h->d_loss);
    if (h->d_hp) hipFree(h->d_hp);
    if (h->h_loss) hipHostFree(h->h_loss);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return CMI_OK;
}

extern "C" int cmi_create(int model, int k, int n_users, int n_items, int n_conds, int device, unsigned flags,
                          cmi_handle *out) {
    if (out) *out = nullptr;
    if (!out || model < 0 || model > CMI_MODEL_CAMF_MCS || k <= 0 || n_users <= 0 || n_items <= 0 || n_conds < 0) {
        g_create_err = "cmi_create: invalid argument";
        return CMI_E_INVALID;
    }
    int ndev = cmi_device_count();
    if (ndev <= 0) {
        g_create_err = "cmi_create: no HIP device visible (libcarskit_mi355x has no CPU fallback)";
        return CMI_E_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        g_create_err = "cmi_create: device index out of range";
        return CMI_E_INVALID;
    }
    if (model == CMI_MODEL_CAMF_C && !(flags & CMI_FLAG_SCHED_SERIAL)) {
        g_create_err =
            "cmi_create: CAMF_C updates the shared condBias vector on every tuple, so its tuples do not commute and "
            "no order-exact level schedule exists; pass CMI_FLAG_SCHED_SERIAL";
        return CMI_E_UNSUPPORTED;
    }
    if (is_ext_model(model) && !(flags & CMI_FLAG_SCHED_SERIAL)) {
        g_create_err =
            "cmi_create: SVD++ / CAMF_ICS / CAMF_LCS / CAMF_MCS update parameters shared by (nearly) every tuple, so no order-exact "
            "parallel schedule exists; pass CMI_FLAG_SCHED_SERIAL";
        return CMI_E_UNSUPPORTED;
    }
    cmi_instance *h = new cmi_instance();
    h->model = model;
    h->k = k;
    h->n_users = n_users;
    h->n_items = n_items;
    h->n_conds = n_conds;
    h->device = device;
    h->flags = flags;
    h->f64 = flags & CMI_FLAG_STATE_F64;
    h->serial = flags & CMI_FLAG_SCHED_SERIAL;
    h->strict = flags & CMI_FLAG_STRICT;
    h->use_graph = !(flags & CMI_FLAG_NO_GRAPH);
    h->want_owner = flags & CMI_FLAG_SCHED_OWNER;
    const char *step = "";
    hipError_t e = hipSuccess;
#define TRY(x)                                                                                          \
    if (e == hipSuccess) {                                                                              \
        step = #x;                                                                                      \
        e = (x);                                                                                        \
    }
    TRY(hipSetDevice(device));
    // experiment builds: CMI_STREAM_CUS="m:r0,r1,..." -- the instance's stream may only use the compute units whose index mod m is one of
    // the residues (the driver deals consecutive mask bits round the XCDs, so m = 8 selects XCDs and m = 16 halves of every XCD)
    if (const char *cus = cmi_exp_env("CMI_STREAM_CUS")) {
        int m = atoi(cus), n_cu = 0;
        uint32_t want = 0, mask[16] = {};
        for (const char *p = strchr(cus, ':'); p && *p; p = strchr(p + 1, ',')) want |= 1u << (atoi(p + 1) & 31);
        TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device));
        for (int c = 0; c < n_cu && c < 512 && m > 0; ++c)
            if (want >> (c % m) & 1) mask[c / 32] |= 1u << (c % 32);
        TRY(hipExtStreamCreateWithCUMask(&h->stream, (uint32_t)((n_cu + 31) / 32), mask));
    } else
        TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    TRY(hipEventCreate(&h->ev0));
    TRY(hipEventCreate(&h->ev1));
    for (int w = 0; w < CMI_STATE_COUNT; ++w) {
        if (!cmi_model_has(model, w)) continue;
        h->state_count[w] = state_elems(h, w);
        size_t bytes = (size_t)h->state_count[w] * esize(h);
        if (bytes == 0) continue;
        TRY(hipMalloc(&h->state[w], bytes));
        TRY(hipMemsetAsync(h->state[w], 0, bytes, h->stream));
    }
    TRY(hipMalloc((void **)&h->d_scratch, 256 * sizeof(double)));
    TRY(hipMalloc((void **)&h->d_loss, sizeof(double)));
    TRY(hipMalloc((void **)&h->d_hp, sizeof(HParams)));
    TRY(hipHostMalloc((void **)&h->h_loss, sizeof(double), hipHostMallocDefault));
    TRY(hipStreamSynchronize(h->stream));
#undef TRY
    if (e != hipSuccess) {
        g_create_err = std::string("cmi_create: ") + step + " failed: " + hipGetErrorString(e);
        cmi_destroy(h);
        return CMI_E_HIP;
    }
    *out = h;
    return CMI_OK;
}

extern "C" int cmi_set_hparams(cmi_handle h, double regU, double regI, double regB, double regC, double global_mean) {
    if (!h) return CMI_E_INVALID;
    h->hp.regU = regU;
    h->hp.regI = regI;
    h->hp.regB = regB;
    h->hp.regC = regC;
    h->hp.gm = h->model == CMI_MODEL_PMF ? 0.0 : global_mean; // PMF.predict is the bare dot product
    return CMI_OK;
}

extern "C" int cmi_set_device_share(cmi_handle h, int instances) {
    if (!h) return CMI_E_INVALID;
    if (instances < 1) CMI_FAIL(h, CMI_E_INVALID, "set_device_share: %d instances", instances);
    h->device_share = instances;
    return CMI_OK;
}

extern "C" int cmi_set_sim_params(cmi_handle h, int num_f, int n_ctx_dims, const int32_t *empty_conds, int n_empty) {
    if (!h) return CMI_E_INVALID;
    if (!is_ext_model(h->model) || h->model == CMI_MODEL_SVDPP) return CMI_OK; // nothing to configure
    if (n_empty < 0 || (n_empty > 0 && !empty_conds) || n_ctx_dims < 1 || (h->model == CMI_MODEL_CAMF_LCS && num_f < 1))
        CMI_FAIL(h, CMI_E_INVALID, "set_sim_params: invalid argument");
    for (int i = 0; i < n_empty; ++i)
        if (empty_conds[i] < 0 || empty_conds[i] >= h->n_conds) CMI_FAIL(h, CMI_E_INVALID, "set_sim_params: empty condition id %d out of range", empty_conds[i]);
    CMI_HIP(h, hipSetDevice(h->device));
    CMI_HIP(h, hipStreamSynchronize(h->stream));
    h->n_ctx_dims = n_ctx_dims;
    h->sim_params_set = true;
    h->empty_conds.assign(empty_conds, empty_conds + n_empty);
    if (h->d_empty) hipFree(h->d_empty);
    h->d_empty = nullptr;
    if (n_empty > 0) {
        CMI_HIP(h, hipMalloc((void **)&h->d_empty, (size_t)n_empty * 4));
        // (never the legacy stream: a synchronous hipMemcpy / hipMemset fails with hipErrorStreamCaptureImplicit while ANOTHER fold's
        //  thread is capturing its level graph -- found by tests/test_gpu_soak.py)
        CMI_HIP(h, hipMemcpyAsync(h->d_empty, empty_conds, (size_t)n_empty * 4, hipMemcpyHostToDevice, h->stream));
        CMI_HIP(h, hipStreamSynchronize(h->stream));
    }
    if (h->model == CMI_MODEL_CAMF_LCS && num_f != h->num_f) {
        if (h->state[CMI_STATE_CF_MATRIX]) hipFree(h->state[CMI_STATE_CF_MATRIX]);
        h->state[CMI_STATE_CF_MATRIX] = nullptr;
        h->num_f = num_f;
        h->state_count[CMI_STATE_CF_MATRIX] = (int64_t)h->n_conds * num_f;
        const size_t bytes = (size_t)h->state_count[CMI_STATE_CF_MATRIX] * esize(h);
        if (bytes) {
            CMI_HIP(h, hipMalloc(&h->state[CMI_STATE_CF_MATRIX], bytes));
            CMI_HIP(h, hipMemsetAsync(h->state[CMI_STATE_CF_MATRIX], 0, bytes, h->stream));
            CMI_HIP(h, hipStreamSynchronize(h->stream));
        }
    }
    return CMI_OK;
}

// ---- state copy-in / copy-back -------------------------------------------------------------------

static int check_state_args(cmi_instance *h, int which, const void *p, int64_t count, int dtype) {
    if (which < 0 || which >= CMI_STATE_COUNT || !p || (dtype != CMI_DTYPE_F32 && dtype != CMI_DTYPE_F64))
        CMI_FAIL(h, CMI_E_INVALID, "state: invalid argument");
    if (!cmi_model_has(h->model, which)) CMI_FAIL(h, CMI_E_INVALID, "state %d does not exist in model %d", which, h->model);
    if (count != h->state_count[which])
        CMI_FAIL(h, CMI_E_INVALID, "state %d: count %lld != expected %lld", which, (long long)count,
                 (long long)h->state_count[which]);
    return CMI_OK;
}

// The multi-GPU exchange keeps a snapshot of the item-side containers as of the last merge (cmi_exchange_setup / _apply) and ships
// `container - snapshot`.  A container rewritten from the host (cmi_set_state, cmi_load_model) restarts from that value on every
// rank, so its snapshot segment follows it -- otherwise the next pack would ship the jump as if it were this rank's SGD move.
static hipError_t refresh_exchange_snapshot(cmi_instance *h, int which) {
    if (!h->d_xsnap) return hipSuccess;
    for (size_t i = 0; i < h->x_which.size(); ++i)
        if (h->x_which[i] == which)
            return hipMemcpyAsync((char *)h->d_xsnap + (size_t)h->x_off[i] * esize(h), h->state[which], (size_t)h->state_count[which] * esize(h),
                                  hipMemcpyDeviceToDevice, h->stream);
    return hipSuccess;
}

extern "C" int cmi_set_state(cmi_handle h, int which, const void *src, int64_t count, int dtype) {
    if (!h) return CMI_E_INVALID;
    if (int rc = check_state_args(h, which, src, count, dtype)) return rc;
    if (count == 0) return CMI_OK;
    if (h->arena_on && which == h->arena_which) { // the whole table is rewritten: the arena's copy of these rows is stale from here on
        h->table_valid = true;
        h->arena_valid = false;
    }
    CMI_HIP(h, hipSetDevice(h->device));
    const bool src_f64 = dtype == CMI_DTYPE_F64;
    if (src_f64 == h->f64) {
        CMI_HIP(h, hipMemcpyAsync(h->state[which], src, (size_t)count * esize(h), hipMemcpyHostToDevice, h->stream));
        CMI_HIP(h, refresh_exchange_snapshot(h, which));
        CMI_HIP(h, hipStreamSynchronize(h->stream));
        return CMI_OK;
    }
    void *stage = nullptr;
    const size_t sb = (size_t)count * (src_f64 ? 8 : 4);
    CMI_HIP(h, hipMalloc(&stage, sb));
    hipError_t e = hipMemcpyAsync(stage, src, sb, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = launch_convert(stage, src_f64, h->state[which], h->f64, count, h->stream);
    if (e == hipSuccess) e = refresh_exchange_snapshot(h, which);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(stage);
    CMI_HIP(h, e);
    return CMI_OK;
}

extern "C" int cmi_get_state(cmi_handle h, int which, void *dst, int64_t count, int dtype) {
    if (!h) return CMI_E_INVALID;
    if (int rc = check_state_args(h, which, dst, count, dtype)) return rc;
    if (count == 0) return CMI_OK;
    if (int rc = cmi_sync_table_from_arena(h)) return rc;
    CMI_HIP(h, hipSetDevice(h->device));
    const bool dst_f64 = dtype == CMI_DTYPE_F64;
    if (dst_f64 == h->f64) {
        CMI_HIP(h, hipMemcpyAsync(dst, h->state[which], (size_t)count * esize(h), hipMemcpyDeviceToHost, h->stream));
        CMI_HIP(h, hipStreamSynchronize(h->stream));
        return CMI_OK;
    }
    void *stage = nullptr;
    const size_t db = (size_t)count * (dst_f64 ? 8 : 4);
    CMI_HIP(h, hipMalloc(&stage, db));
    hipError_t e = launch_convert(h->state[which], h->f64, stage, dst_f64, count, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dst, stage, db, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(stage);
    CMI_HIP(h, e);
    return CMI_OK;
}

extern "C" int cmi_state_device_ptr(cmi_handle h, int which, void **ptr, int64_t *count, int *dtype) {
    if (!h || which < 0 || which >= CMI_STATE_COUNT) return CMI_E_INVALID;
    if (!cmi_model_has(h->model, which)) CMI_FAIL(h, CMI_E_INVALID, "state %d does not exist in model %d", which, h->model);
    if (int rc = cmi_sync_table_from_arena(h)) return rc;
    // The header documents this pointer for the host's in-place epoch-boundary exchange, i.e. the host may WRITE the table.  With
    // the spoke arena holding the live rows of this container, such a write would be lost (the next epoch reads the arena and the
    // following gather overwrites the table), so handing the pointer out makes the table the master copy: the next epoch
    // re-scatters it into the arena (one streaming pass; only paid by hosts that ask for the pointer of the arena's container).
    if (h->arena_on && which == h->arena_which) h->arena_valid = false;
    if (ptr) *ptr = h->state[which];
    if (count) *count = h->state_count[which];
    if (dtype) *dtype = h->f64 ? CMI_DTYPE_F64 : CMI_DTYPE_F32;
    return CMI_OK;
}

// ---- ratings + schedule ----------------------------------------------------------------------------

template <typename V>
static hipError_t upload(void **dst, const std::vector<V> &v, hipStream_t s) {
    *dst = nullptr;
    if (v.empty()) return hipSuccess;
    hipError_t e = hipMalloc(dst, v.size() * sizeof(V));
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(*dst, v.data(), v.size() * sizeof(V), hipMemcpyHostToDevice, s);
}

template <typename V>
static hipError_t upload(void **dst, const V *v, size_t count, hipStream_t s) {
    *dst = nullptr;
    if (!count) return hipSuccess;
    hipError_t e = hipMalloc(dst, count * sizeof(V));
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(*dst, v, count * sizeof(V), hipMemcpyHostToDevice, s);
}
// a host array that is written once, in ranges, before it is read: not zero-filled first (5.6 GB for the north_star tuple stream)
template <typename V>
struct HostBuf {
    std::unique_ptr<V[]> p;
    size_t n = 0;
    explicit HostBuf(size_t count = 0) : p(count ? new V[count] : nullptr), n(count) {}
    V *data() { return p.get(); }
    const V *data() const { return p.get(); }
    V &operator[](size_t i) { return p[i]; }
    const V &operator[](size_t i) const { return p[i]; }
};
template <typename V>
static hipError_t upload(void **dst, const HostBuf<V> &v, hipStream_t s) {
    return upload(dst, v.data(), v.n, s);
}

// Spoke arena bookkeeping: next[p] = stream position of the next tuple of the same spoke row (a row's tuples sit in ascending levels,
// hence ascending positions); the row's last tuple wraps to its first -- that is where the row waits for the next epoch.
// first[row] = position of the row's first tuple, -1 for a row without tuples.
static void arena_positions(int64_t n, const int32_t *spoke, int64_t n_spokes, int32_t *next, int32_t *first) {
    const int nt = host_threads(n);
    if (n < ((int64_t)1 << 22) || nt < 2) { // small: one backward walk
        for (int64_t r = 0; r < n_spokes; ++r) first[r] = -1;
        for (int64_t p = n - 1; p >= 0; --p) {
            const int32_t r = spoke[p];
            next[p] = first[r]; // -1 for the row's last tuple: patched below
            first[r] = (int32_t)p;
        }
        for (int64_t p = 0; p < n; ++p)
            if (next[p] < 0) next[p] = first[spoke[p]];
        return;
    }
    // Large: the backward walk's state is one entry per spoke row, and the rows do not interact -- so the rows are split into BUCKETS
    // of 2^sh consecutive ids whose entries fit a core's cache, and the walk runs per bucket.  (1) ranges of positions append
    // (position, row) to one list per (range, bucket); (2) every bucket walks its lists backwards (ranges descending, a list from its
    // end), leaves the answer in the list entry and patches the rows' last tuples once its rows' first positions are final; (3) every
    // range of positions copies its lists' answers into its own part of `next` -- no two threads ever write the same cache line.
    int sh = 0;
    while (((n_spokes - 1) >> sh) >= 256) ++sh;
    const int nbk = (int)((n_spokes - 1) >> sh) + 1;
    struct PR {
        int32_t p, r; // position and row; `r` becomes the answer in step (2)
    };
    std::vector<std::vector<PR>> lists((size_t)nt * (size_t)nbk);
    std::vector<int64_t> lo((size_t)nt + 1, n);
    const int64_t step = (n + nt - 1) / nt;
    for (int t = 0; t <= nt; ++t) lo[(size_t)t] = std::min<int64_t>(n, (int64_t)t * step);
    parallel_ranges(nt, nt, [&](int, int64_t t0, int64_t t1) {
        for (int64_t t = t0; t < t1; ++t) {
            std::vector<PR> *L = &lists[(size_t)t * (size_t)nbk];
            const int64_t b = lo[(size_t)t], e = lo[(size_t)t + 1];
            for (int k = 0; k < nbk; ++k) L[k].reserve((size_t)((e - b) / nbk + (e - b) / (4 * nbk) + 16));
            for (int64_t p = b; p < e; ++p) L[spoke[p] >> sh].push_back(PR{(int32_t)p, spoke[p]});
        }
    });
    parallel_ranges(nbk, nt, [&](int, int64_t k0, int64_t k1) {
        for (int64_t k = k0; k < k1; ++k) {
            const int64_t r0 = k << sh, r1 = std::min<int64_t>(n_spokes, (k + 1) << sh);
            for (int64_t r = r0; r < r1; ++r) first[r] = -1;
            for (int t = nt - 1; t >= 0; --t) {
                std::vector<PR> &L = lists[(size_t)t * (size_t)nbk + (size_t)k];
                for (size_t x = L.size(); x-- > 0;) {
                    const int32_t r = L[x].r;
                    L[x].r = first[r]; // -1 for the row's last tuple
                    first[r] = L[x].p;
                }
            }
            for (int t = 0; t < nt; ++t) // the rows' last tuples wrap to their first
                for (PR &x : lists[(size_t)t * (size_t)nbk + (size_t)k])
                    if (x.r < 0) x.r = first[spoke[x.p]];
        }
    });
    parallel_ranges(nt, nt, [&](int, int64_t t0, int64_t t1) {
        for (int64_t t = t0; t < t1; ++t)
            for (int k = 0; k < nbk; ++k)
                for (const PR &x : lists[(size_t)t * (size_t)nbk + (size_t)k]) next[x.p] = x.r;
    });
}
// host-only export of the same (tests/test_chain_schedule.py)
extern "C" int cmi_arena_positions(int64_t n, const int32_t *spoke, int32_t n_spokes, int32_t *next, int32_t *first) {
    if (n < 0 || n_spokes < 0 || (n > 0 && (!spoke || !next)) || (n_spokes > 0 && !first)) return CMI_E_INVALID;
    for (int64_t p = 0; p < n; ++p)
        if (spoke[p] < 0 || spoke[p] >= n_spokes) return CMI_E_INVALID;
    arena_positions(n, spoke, n_spokes, next, first);
    return CMI_OK;
}

// Hub-chain level schedule (level_schedule.cpp): used when forced, or when its levels are wide enough to fill the chip
// (narrow levels -- heavy-tailed degrees, tiny data -- stay with the plain levels and their narrow-run launches).
static int chain_max_len() {
    int v = 16; // the kernel stages a unit's ids in 16 LDS slots per group
    if (const char *env = getenv("CMI_CHAIN_MAX")) v = atoi(env);
    return v < 1 ? 1 : (v > 16 ? 16 : v);
}
static void free_keep(ChainDeviceKeep &keep) { keep.release(); } // (early: 12 bytes per tuple; the destructor covers every other exit)
static bool try_chain(cmi_instance *h, int64_t n, const int32_t *u, const int32_t *j, ChainSchedule &csch, ChainDeviceKeep &keep) {
    // the side that carries the context-bias rows is preferred as the hub side (CAMF_CI: items, CAMF_CU: users): measured on the C5
    // share (CAMF_CU k=256, 128 conditions) 50.1 ms per epoch along users (36.4 M units) against 64.3 ms along items (31.9 M units)
    int hub = h->model == CMI_MODEL_CAMF_CU ? -2 : (h->model == CMI_MODEL_CAMF_CI ? -3 : -1);
    if (const char *env = getenv("CMI_CHAIN_HUB")) hub = !strcmp(env, "item") ? 1 : (!strcmp(env, "user") ? 0 : hub);
    // large sets: the schedule is built on the device (sched_device.hip: the same schedule, element for element; the host walk is
    // 11 ns per tuple on one core).  CMI_SCHEDULE_DEVICE=0 / 1: never / at any size.
    int64_t dev_min = (int64_t)1 << 21;
    if (const char *env = getenv("CMI_SCHEDULE_DEVICE")) dev_min = atoi(env) ? 0 : INT64_MAX;
    bool built = false;
    if (n >= dev_min) {
        built = build_chain_schedule_device(h->device, h->stream, n, u, j, h->n_users, h->n_items, hub, chain_max_len(), csch, &keep);
        if (!built) {
            (void)hipGetLastError();
            free_keep(keep);
        }
    }
    if (!built && !build_chain_schedule(n, u, j, h->n_users, h->n_items, hub, chain_max_len(), csch)) return false;
    int64_t min_width = 2048; // mean units per level
    if (const char *env = cmi_exp_env("CMI_CHAIN_MIN_WIDTH")) min_width = atoll(env);
    const bool forced = h->flags & CMI_FLAG_SCHED_CHAIN;
    // Narrow levels (heavy-tailed degrees, tiny data) keep the plain levels and their narrow-run launches: measured on C3-size
    // Zipf(0.8) items, the chain schedule has 2.5x fewer levels (434 K vs 1.10 M) but a narrow chain level is latency-bound on
    // the HBM round trip of EVERY spoke row of its longest unit (one row in flight per group), 3.77 s per epoch against 2.61 s for
    // the plain narrow-run walk.  Forced (CMI_FLAG_SCHED_CHAIN) it still runs, one launch per level.
    if (!forced && csch.n_units() < min_width * csch.n_levels()) {
        free_keep(keep);
        return false;
    }
    h->chain = true;
    h->chain_hub_item = csch.hub_is_item != 0;
    h->n_units = csch.n_units();
    return true;
}

static int set_ratings_impl(cmi_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, const double *r, int32_t n_ctx,
                            const int32_t *ctx_ptr, const int32_t *ctx_conds);

// the exception barrier of the boundary: the schedule construction allocates O(tuples) host memory on the host pool's threads
// (host_pool.hpp hands a range body's exception to the caller); nothing C++ may cross into a C / JNI / ctypes host
extern "C" int cmi_set_ratings(cmi_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx,
                               const double *r, int32_t n_ctx, const int32_t *ctx_ptr, const int32_t *ctx_conds) {
    if (!h) return CMI_E_INVALID;
    try {
        return set_ratings_impl(h, n, u, j, ctx, r, n_ctx, ctx_ptr, ctx_conds);
    } catch (const std::exception &e) {
        h->have_ratings = false;
        CMI_FAIL(h, CMI_E_HOST, "set_ratings: host-side failure: %s", e.what());
    } catch (...) {
        h->have_ratings = false;
        CMI_FAIL(h, CMI_E_HOST, "set_ratings: host-side failure (unknown exception)");
    }
}

static int set_ratings_impl(cmi_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, const double *r, int32_t n_ctx,
                            const int32_t *ctx_ptr, const int32_t *ctx_conds) {
    const bool contextual = !is_2d_model(h->model);
    if (n < 0 || (n > 0 && (!u || !j || !r))) CMI_FAIL(h, CMI_E_INVALID, "set_ratings: null tuple arrays");
    if (is_ext_model(h->model) && h->model != CMI_MODEL_SVDPP && (!h->sim_params_set || (h->model == CMI_MODEL_CAMF_LCS && h->num_f < 1)))
        CMI_FAIL(h, CMI_E_INVALID, "set_ratings: call cmi_set_sim_params first (EmptyContextConditions%s)", h->model == CMI_MODEL_CAMF_LCS ? ", numF" : "");
    if (contextual && is_ext_model(h->model) && n_ctx > 0 && ctx_ptr) {
        for (int32_t c = 0; c < n_ctx; ++c)
            if (ctx_ptr[c + 1] - ctx_ptr[c] > 16) CMI_FAIL(h, CMI_E_UNSUPPORTED, "set_ratings: more than 16 conditions per context");
        // condition i of a context is paired with EmptyContextConditions.get(i) (CAMF_ICS.java:56,88: an IndexOutOfBoundsException in the
        // reference when the data has fewer ':na' conditions than a context has conditions, e.g. Frappe, which has none)
        for (int32_t c = 0; c < n_ctx; ++c)
            if (ctx_ptr[c + 1] - ctx_ptr[c] > (int32_t)h->empty_conds.size())
                CMI_FAIL(h, CMI_E_INVALID, "set_ratings: context %d has %d conditions but EmptyContextConditions has %d entries "
                         "(IndexOutOfBoundsException in the reference, CAMF_ICS.java:88)", c, ctx_ptr[c + 1] - ctx_ptr[c], (int)h->empty_conds.size());
    }
    if (contextual && (n_ctx < 0 || !ctx_ptr || (n > 0 && !ctx) || (n_ctx > 0 && ctx_ptr[n_ctx] > 0 && !ctx_conds)))
        CMI_FAIL(h, CMI_E_INVALID, "set_ratings: context table required for model %d", h->model);
    CMI_HIP(h, hipSetDevice(h->device));
    CMI_HIP(h, hipStreamSynchronize(h->stream));
    free_ratings(h);

    const bool times = getenv("CMI_SETUP_TIMES") != nullptr; // per-phase wall times of this call on stderr
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *w) {
        if (!times) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "set_ratings %s %.3f s\n", w, std::chrono::duration<double>(t - T0).count());
        T0 = t;
    };
    // validate ids (the reference would throw ArrayIndexOutOfBounds inside the loop)
    int dmax = 0;
    if (contextual) {
        if (ctx_ptr[0] != 0) CMI_FAIL(h, CMI_E_INVALID, "set_ratings: ctx_ptr[0] != 0");
        for (int32_t c = 0; c < n_ctx; ++c) {
            const int32_t len = ctx_ptr[c + 1] - ctx_ptr[c];
            if (len < 0) CMI_FAIL(h, CMI_E_INVALID, "set_ratings: ctx_ptr not monotone at %d", c);
            if (len > dmax) dmax = len;
        }
        for (int32_t q = 0; q < ctx_ptr[n_ctx]; ++q)
            if (ctx_conds[q] < 0 || ctx_conds[q] >= h->n_conds)
                CMI_FAIL(h, CMI_E_INVALID, "set_ratings: condition id %d out of range [0,%d)", ctx_conds[q], h->n_conds);
    }
    {
        const int nt = host_threads(n);
        std::vector<int64_t> bad((size_t)nt, -1); // first offending tuple of every range; the earliest is reported, as a sequential scan would
        parallel_ranges(n, nt, [&](int part, int64_t b, int64_t e) {
            for (int64_t t = b; t < e; ++t)
                if (u[t] < 0 || u[t] >= h->n_users || j[t] < 0 || j[t] >= h->n_items || (contextual && (ctx[t] < 0 || ctx[t] >= n_ctx))) {
                    bad[(size_t)part] = t;
                    return;
                }
        });
        for (int64_t t : bad) {
            if (t < 0) continue;
            if (u[t] < 0 || u[t] >= h->n_users) CMI_FAIL(h, CMI_E_INVALID, "set_ratings: user id %d out of range at tuple %lld", u[t], (long long)t);
            if (j[t] < 0 || j[t] >= h->n_items) CMI_FAIL(h, CMI_E_INVALID, "set_ratings: item id %d out of range at tuple %lld", j[t], (long long)t);
            CMI_FAIL(h, CMI_E_INVALID, "set_ratings: context id %d out of range at tuple %lld", ctx[t], (long long)t);
        }
    }
    if (n >= ((int64_t)1 << 31) - 1024) CMI_FAIL(h, CMI_E_UNSUPPORTED, "set_ratings: more than 2^31-1025 tuples per instance");
    lap("validate");

    h->n = n;
    h->n_ctx = contextual ? n_ctx : 0;
    h->dmax = dmax;
    LaunchCfg cfg{h->model, h->strict};
    h->fast = !h->serial && has_fast_path(h->k, dmax, h->f64, cfg);
    h->small = !h->serial && !h->fast && has_small_path(h->k, dmax, h->f64, cfg);

    // schedule
    LevelSchedule sch;
    ChainSchedule csch;
    OwnerSchedule osch;
    const bool chain_ok = !h->serial && !h->want_owner && !(h->flags & CMI_FLAG_NO_CHAIN) &&
                          has_chain_path(h->model, h->k, dmax, h->n_conds, h->f64, h->strict) && !cmi_exp_env("CMI_NO_CHAIN");
    if ((h->flags & CMI_FLAG_SCHED_CHAIN) && !chain_ok)
        CMI_FAIL(h, CMI_E_UNSUPPORTED, "set_ratings: CMI_FLAG_SCHED_CHAIN: no hub-chain kernel for model %d, k=%d, %s state%s (or another "
                 "schedule flag is set)", h->model, h->k, h->f64 ? "fp64" : "fp32", h->strict ? ", strict" : "");
    h->owner = false;
    h->owner_stalled = false;
    if (h->want_owner && (h->serial || !has_owner_path(h->model, h->k, h->n_conds, h->f64, h->strict)))
        CMI_FAIL(h, CMI_E_UNSUPPORTED, "set_ratings: CMI_FLAG_SCHED_OWNER: no owner kernel for model %d, k=%d, %d conditions, %s state%s (or "
                 "another schedule flag is set)", h->model, h->k, h->n_conds, h->f64 ? "fp64" : "fp32", h->strict ? ", strict" : "");
    // the hub-chain levels first: wide data is theirs
    // The spoke arena of a large set is tens of GB (north_star: 102 GB).  hipMalloc hands out CLEAN device memory at once (0.001 s for
    // 102 GB) but waits for the driver's wipe of blocks this process freed a moment ago (~30 ms per GB: tools/exp/hipmalloc_probe.py,
    // tools/micro/alloc_time.hip), and the schedule build below frees GBs of temporaries.  So when the sizes say an arena is likely (a
    // side's table >= 2 GiB, or it is forced) it is allocated HERE, before anything of this call has been freed; the arena block below
    // takes it over or frees it.  (Round 5 first allocated it on a side thread beside the schedule build: 1.9 s for the same call, and
    // every other HIP call of the build slowed down beside it.)
    struct SpecArena {
        void *ptr = nullptr;
        size_t bytes = 0;
        void *take(size_t want) {
            if (ptr && bytes == want) {
                void *p = ptr;
                ptr = nullptr;
                return p;
            }
            return nullptr;
        }
        ~SpecArena() {
            if (ptr) (void)hipFree(ptr);
        }
    } spec;
    if (!h->serial && chain_ok && n >= ((int64_t)1 << 21) && !(h->flags & CMI_FLAG_NO_ARENA) && !cmi_exp_env("CMI_NO_ARENA") && h->k >= 64 &&
        h->k % (h->f64 ? 2 : 4) == 0) {
        const size_t row = (size_t)h->k * esize(h), arena_bytes = (size_t)n * row;
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        const bool likely = (h->flags & CMI_FLAG_SPOKE_ARENA) || cmi_exp_env("CMI_ARENA") || (size_t)std::max(h->n_users, h->n_items) * row >= ((size_t)2 << 30);
        if (likely && (double)arena_bytes <= 0.6 * (double)free_b) {
            if (hipMalloc(&spec.ptr, arena_bytes) == hipSuccess) spec.bytes = arena_bytes;
            else {
                spec.ptr = nullptr;
                (void)hipGetLastError(); // the arena block decides again (and reports) with what is free then
            }
        }
    }
    lap("arena allocation");
    ChainDeviceKeep keep; // device-built schedule: the tuple ids and the permutation stay on the device for the stream build
    const bool use_chain = !h->serial && chain_ok && n > 0 && try_chain(h, n, u, j, csch, keep);
    const bool dev_stream = use_chain && keep.d_perm != nullptr;
    bool use_owner = h->want_owner;
    h->sched_note.clear();
    if (!use_owner && !use_chain && !h->serial && n >= ((int64_t)1 << 16) &&
        !has_owner_path(h->model, h->k, h->n_conds, h->f64, h->strict) && !(h->flags & CMI_FLAG_NO_OWNER))
        h->sched_note = "narrow dependency levels on a large data set (heavy-tailed degrees?) and no owner kernel for this configuration "
                        "(limits: <= 384 conditions, k <= 256 (fp64: 128)): the level walk runs -- order-exact, roughly 10x slower on such data";
    if (!use_owner && !use_chain && !h->serial && !(h->flags & (CMI_FLAG_SCHED_CHAIN | CMI_FLAG_NO_OWNER)) &&
        !cmi_exp_env("CMI_NO_OWNER") && has_owner_path(h->model, h->k, h->n_conds, h->f64, h->strict)) {
        // Narrow levels on a large data set = heavy-tailed degrees: every level costs a kernel boundary or a workgroup barrier (>= 2 us),
        // and there are at least as many levels as the hottest row has tuples.  The owner epoch pays ~0.3 us per tuple of the hottest
        // row it owns and a hand-off (0.86 us measured, tools/exp_owner_handoff.py) per tuple of the hottest row on the other side,
        // plus ~0.3 ms for the tag / untag passes and the launch.  Taken for heavy-tailed data when that is at least twice faster
        // (measured: 3.3-3.7x at 100 K - 800 K ratings with Zipf(1.1) items, tests/tools/bench_zipf_small.py).
        int64_t min_tuples = (int64_t)1 << 16;
        if (const char *env = cmi_exp_env("CMI_OWNER_MIN_TUPLES")) min_tuples = atoll(env);
        if (n >= min_tuples) {
            std::vector<int32_t> du((size_t)h->n_users, 0), dj((size_t)h->n_items, 0);
            for (int64_t t = 0; t < n; ++t) {
                du[(size_t)u[t]]++;
                dj[(size_t)j[t]]++;
            }
            const double mu = *std::max_element(du.begin(), du.end()), mj = *std::max_element(dj.begin(), dj.end());
            const bool hub_item = mj >= mu; // what build_owner_schedule will pick
            int64_t rows_used = 0;          // rows of the hub side that have tuples at all
            for (int32_t d : (hub_item ? dj : du)) rows_used += d > 0;
            // heavy-tailed = the hottest hub row holds at least 8 average rows' worth of tuples (uniform data: 1.5 - 2)
            const bool skewed = std::max(mu, mj) * (double)rows_used >= 8.0 * (double)n;
            // the tagged record table is addressed through one buffer resource: below 4 GB
            const int64_t rec_bytes = owner_record_stride(h->model, h->k, h->n_conds, h->f64, hub_item) * 8;
            const bool table_fits = ((int64_t)(hub_item ? h->n_users : h->n_items) + 4096) * rec_bytes < ((int64_t)1 << 32) - 65536;
            h->sched_note.clear();
            if (skewed && !table_fits)
                h->sched_note = "heavy-tailed degrees, but the owner epoch's tagged record table would exceed 4 GB (too many spoke rows at this k): "
                                "the level walk runs instead -- order-exact, roughly 10x slower on such data";
            if (skewed && table_fits) {
                // owner: the hottest chains, or the bulk spread over ~1 000 owners at ~0.5 us per tuple of a list that switches rows
                const double est_owner = std::max(std::max(std::max(mu, mj) * 0.3e-6, std::min(mu, mj) * 1e-6), (double)n * 0.5e-6 / 1024.0) + 0.3e-3;
                // levels: a boundary or barrier per level plus the traffic (2 KB per tuple at ~5 TB/s)
                const double est_levels = (double)count_plain_levels(n, u, j, h->n_users, h->n_items) * 2e-6 + (double)n * 0.4e-9;
                use_owner = est_levels >= 2.0 * est_owner;
            }
        }
    }
    if (use_owner) {
        int hub = -1;
        if (const char *env = getenv("CMI_OWNER_HUB")) hub = !strcmp(env, "item") ? 1 : (!strcmp(env, "user") ? 0 : -1);
        if (n > 0) {
            // the owners must all be resident: as many as the device holds wavefronts of the kernel (either hub side: same registers)
            int waves = owner_grid_waves(h->device, h->model, h->n_conds, h->k, h->f64, true);
            // cmi_set_device_share: this instance's share of the resident wavefronts (whole workgroups of four owners), so that the
            // persistent launches of the instances that train side by side are co-resident
            if (h->device_share > 1 && waves > 0) waves = std::max(8, waves / h->device_share / 4 * 4);
            if (const char *env = getenv("CMI_OWNER_WAVES")) {
                const int v = atoi(env);
                if (v >= 1 && v < waves) waves = v;
            }
            if (waves < 1) CMI_FAIL(h, CMI_E_HIP, "set_ratings: owner kernel occupancy query failed");
            if (!build_owner_schedule(n, u, j, h->n_users, h->n_items, hub, waves, owner_depth(), osch))
                CMI_FAIL(h, CMI_E_UNSUPPORTED, "set_ratings: owner schedule construction failed");
            // Teams: the leading owners (the hottest rows, alone in their lists) whose list is long enough to bound the epoch run as three
            // wavefronts in a workgroup of their own instead of one wavefront among four.  A team takes a whole workgroup, so the owner
            // count shrinks by three per team and the rows are dealt again.
            h->n_team = 0;
            int64_t team_min = 8192;
            const char *team_env = getenv("CMI_OWNER_TEAM");
            if (const char *env = getenv("CMI_OWNER_TEAM_MIN")) team_min = atoll(env);
            auto leading_single_hub = [&](const OwnerSchedule &os, int limit) {
                int t = 0;
                for (; t < limit && t < (int)os.n_owners(); ++t) {
                    const int64_t b = os.own_off[(size_t)t], e = os.own_off[(size_t)t + 1];
                    if (e - b < team_min) break;
                    bool single = true;
                    for (int64_t pos = b + 1; pos < e && single; ++pos) single = os.flags[(size_t)pos] & OWN_HUB_FWD;
                    if (!single) break;
                }
                return t;
            };
            const int wgs = (waves + 3) / 4; // resident workgroups
            if (h->strict || (team_env && !strcmp(team_env, "0"))) {
                // one wavefront per owner throughout (the strict form has no team body; CMI_OWNER_TEAM=0: tests and A/B runs).
                // (Round 5 also refused teams to sharing instances and to fp64 at k <= 64 because that form was measured inexact beside
                // another owner epoch.  The cause was a store-data hazard in the record stores of BOTH bodies -- the compiler had merely
                // happened to reuse the data registers at once only in that instantiation -- fixed in owner_kernels.hip owner_st_words;
                // tests/test_gpu_soak.py holds every instantiation beside other owner epochs bit for bit.  docs/history/r06.md 1.)
            } else if (team_env && !strcmp(team_env, "all")) { // testing: every owner a team, whatever its list
                if (waves > wgs) {
                    waves = wgs;
                    if (!build_owner_schedule(n, u, j, h->n_users, h->n_items, hub, waves, owner_depth(), osch))
                        CMI_FAIL(h, CMI_E_UNSUPPORTED, "set_ratings: owner schedule construction failed");
                }
                h->n_team = waves;
            } else {
                int t = leading_single_hub(osch, wgs / 2);
                if (t > 0) {
                    const int fewer = waves - 3 * t;
                    if (fewer >= t + 1 && build_owner_schedule(n, u, j, h->n_users, h->n_items, hub, fewer, owner_depth(), osch)) {
                        waves = fewer;
                        t = std::min(t, leading_single_hub(osch, t));
                    } else {
                        t = 0;
                        if (!build_owner_schedule(n, u, j, h->n_users, h->n_items, hub, waves, owner_depth(), osch))
                            CMI_FAIL(h, CMI_E_UNSUPPORTED, "set_ratings: owner schedule construction failed");
                    }
                }
                h->n_team = t;
            }
            h->owner = true;
            h->owner_hub_item = osch.hub_is_item != 0;
            h->n_owners = waves;
        }
    }
    if (h->owner) {
        h->level_off = {0, n};
        h->max_level = osch.max_load;
        h->slot_off = {0, (int64_t)h->n_owners};
        h->n_slots = h->n_owners;
        h->sched_levels = 1;
        sch.perm.swap(osch.perm);
    } else {
        if (h->serial) {
            sch.level_off = {0, n};
            sch.max_level = n;
            // CAMF_C: conflict-free CRS blocks (consecutive tuples sharing no user and no item, <= 64) -- see sgd_camfc_blocks
            h->blk_off.clear();
            if (h->model == CMI_MODEL_CAMF_C && !h->strict && h->k <= 256 && dmax <= 16 && n > 0 && n < ((int64_t)1 << 31) &&
                camfc_blocks_lds(h->n_conds, dmax, esize(h)) <= 64 * 1024 && !getenv("CMI_NO_CAMFC_BLOCKS")) {
                std::vector<int32_t> off;
                build_conflict_free_blocks(n, u, j, h->n_users, h->n_items, 64, off);
                // shorter runs: the serial wave is faster -- the pipelined one (camfc_pipe.hip, 0.55-0.63 us per tuple at any run length)
                // breaks even with the block kernel (about 1.3 us per block + 0.28 us per tuple) near 5 tuples per block, the one-ahead
                // wave near 3 (tests/tools/bench_camfc_paths.py)
                const double min_run = camfc_pipe_supported(h->k, h->n_conds, dmax) ? 5.0 : 3.0;
                if ((double)n / (double)(off.size() - 1) >= min_run) h->blk_off.swap(off);
            }
        } else if (use_chain) {
            // hub-chain levels: level_off indexes UNITS; sch.perm carries the stream order
            sch.perm.swap(csch.perm);
            sch.level_off = csch.level_off;
            sch.max_level = csch.max_level_units;
        } else {
            int order = LEVEL_ORDER_CRS;
            if (const char *env = getenv("CMI_LEVEL_ORDER")) {
                if (!strcmp(env, "item")) order = LEVEL_ORDER_ITEM;
                else if (!strcmp(env, "user")) order = LEVEL_ORDER_USER;
                else if (!strcmp(env, "xcd")) order = LEVEL_ORDER_XCD;
            }
            if (!build_level_schedule(n, u, j, h->n_users, h->n_items, order, sch))
                CMI_FAIL(h, CMI_E_UNSUPPORTED, "set_ratings: schedule construction failed");
        }
#ifdef CMI_TIMING_EXPERIMENTS // never in a release build: merging dependent levels gives WRONG results (boundary-cost timing only)
        if (const char *env = cmi_exp_env("CMI_DEBUG_MERGE_LEVELS")) {
            const int m = atoi(env);
            if (m > 1) {
                fprintf(stderr, "[cmi] CMI_DEBUG_MERGE_LEVELS=%d: dependent levels merged, results are WRONG (timing experiment)\n", m);
                std::vector<int64_t> lo;
                for (size_t i = 0; i + 1 < sch.level_off.size(); i += (size_t)m) lo.push_back(sch.level_off[i]);
                lo.push_back(sch.level_off.back());
                sch.level_off = lo;
            }
        }
#endif
        h->level_off = sch.level_off;
        h->max_level = sch.max_level;
        const int64_t n_levels = (int64_t)h->level_off.size() - 1;
        h->sched_levels = n_levels;
        // narrow runs (heavy-tailed degrees): every maximal run of >= TAIL_MIN_LEVELS consecutive levels with <= TAIL_MAX
        // tuples each is walked by ONE single-workgroup launch instead of one launch per level
        h->tail_len.assign((size_t)n_levels, 0);
        h->n_launches = 0;
        if (!h->serial && !h->chain && !cmi_exp_env("CMI_NO_TAIL")) build_narrow_runs(h->level_off, 256, 16, h->tail_len);
        h->slot_off.assign((size_t)n_levels + 1, 0);
        for (int64_t l = 0; l < n_levels; ++l) {
            const int cnt = (int)(h->level_off[(size_t)l + 1] - h->level_off[(size_t)l]);
            int blocks = h->serial ? 0
                         : h->chain ? chain_level_blocks(h->k, dmax, h->f64, cnt)
                         : h->fast ? level_blocks_f32_fast(h->k, cnt)
                         : h->small ? level_blocks_small(h->k, dmax, cnt)
                                    : level_blocks_generic(cnt);
            if (h->tail_len[(size_t)l] > 0) blocks = 1;       // a narrow run owns one loss slot ...
            else if (h->tail_len[(size_t)l] < 0) blocks = 0;  // ... at its first level
            if (h->tail_len[(size_t)l] >= 0) ++h->n_launches;
            h->slot_off[(size_t)l + 1] = h->slot_off[(size_t)l] + blocks;
        }
        h->n_slots = h->slot_off[(size_t)n_levels];
        h->n_tail = 0;
        for (int32_t t : h->tail_len)
            if (t > 0) h->n_tail += t;
    }

    lap("schedule");
    // tuple stream in schedule order, conditions pre-expanded to [n x dmax] (-1 padded) so the kernels
    // need no ctx -> condition-list indirection.
    const int64_t ns = n;
    const int64_t hs = dev_stream ? 0 : ns; // (the device-built schedule builds the stream on the device too: no host copy of it)
    HostBuf<int32_t> su((size_t)hs), sj((size_t)hs), sconds((size_t)hs * (size_t)dmax);
    HostBuf<float> sr32(h->f64 ? 0 : (size_t)hs);
    HostBuf<double> sr64(h->f64 ? (size_t)hs : 0);
    // every stream slot is a gather through the schedule's permutation: ranges of slots on the host's cores, the tuple arrays' entries
    // requested 16 slots ahead
    if (!dev_stream)
    parallel_ranges(ns, host_threads(ns), [&](int, int64_t s0, int64_t s1) {
        for (int64_t s = s0; s < s1; ++s) {
            if (!h->serial && s + 16 < s1) {
                const int64_t tp = sch.perm[(size_t)s + 16];
                if (tp >= 0) {
                    __builtin_prefetch(&u[tp]);
                    __builtin_prefetch(&j[tp]);
                    __builtin_prefetch(&r[tp]);
                    if (dmax > 0) __builtin_prefetch(&ctx[tp]);
                }
            }
            const int64_t t = h->serial ? s : sch.perm[(size_t)s];
            int32_t *row = dmax > 0 ? &sconds[(size_t)s * (size_t)dmax] : nullptr;
            if (t < 0) { // padding slot
                su[(size_t)s] = -1;
                sj[(size_t)s] = 0;
                if (h->f64) sr64[(size_t)s] = 0.0;
                else sr32[(size_t)s] = 0.f;
                for (int d = 0; d < dmax; ++d) row[d] = -1;
                continue;
            }
            su[(size_t)s] = u[t];
            sj[(size_t)s] = j[t];
            if (h->f64) sr64[(size_t)s] = r[t];
            else sr32[(size_t)s] = (float)r[t];
            if (dmax > 0) {
                const int32_t b = ctx_ptr[ctx[t]], e = ctx_ptr[ctx[t] + 1];
                int d = 0;
                for (int32_t q = b; q < e; ++q) row[d++] = ctx_conds[q];
                for (; d < dmax; ++d) row[d] = -1;
            }
        }
    });
    lap("stream");
    hipError_t e = hipSuccess;
    if (!dev_stream) {
        e = upload((void **)&h->d_su, su, h->stream);
        if (e == hipSuccess) e = upload((void **)&h->d_sj, sj, h->stream);
        if (e == hipSuccess) e = upload((void **)&h->d_sconds, sconds, h->stream);
        if (e == hipSuccess) e = h->f64 ? upload(&h->d_sr, sr64, h->stream) : upload(&h->d_sr, sr32, h->stream);
    }
    if (e == hipSuccess && contextual) {
        std::vector<int32_t> cp(ctx_ptr, ctx_ptr + n_ctx + 1), cc(ctx_conds, ctx_conds + ctx_ptr[n_ctx]);
        h->ctx_nnz = (int64_t)cc.size();
        e = upload((void **)&h->d_ctx_ptr, cp, h->stream);
        if (e == hipSuccess) e = upload((void **)&h->d_ctx_conds, cc, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream); // cp/cc are locals
    }
    if (e == hipSuccess && dev_stream) {
        e = hipMalloc((void **)&h->d_su, (size_t)ns * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&h->d_sj, (size_t)ns * 4);
        if (e == hipSuccess && dmax > 0) e = hipMalloc((void **)&h->d_sconds, (size_t)ns * (size_t)dmax * 4);
        if (e == hipSuccess) e = hipMalloc(&h->d_sr, (size_t)ns * esize(h));
        if (e == hipSuccess)
            e = (hipError_t)stream_build_device(h->stream, ns, keep.d_u, keep.d_j, keep.d_perm, ctx, r, h->d_ctx_ptr, h->d_ctx_conds, dmax, h->f64, h->d_su,
                                                h->d_sj, h->d_sconds, h->d_sr);
    }
    free_keep(keep);
    if (e == hipSuccess && h->chain) e = upload((void **)&h->d_unit_off, csch.unit_off, h->stream);
    lap("uploads");
    if (e == hipSuccess && h->chain && !(h->flags & CMI_FLAG_NO_ARENA) && !cmi_exp_env("CMI_NO_ARENA") && h->k >= 64 && h->k % (h->f64 ? 2 : 4) == 0) {
        // Spoke arena (SgdArgs::arena): worth its memory when the spoke table is large -- random 512-B rows over >= 2 GiB run at ~0.5 of
        // the HBM peak (address-translation misses, DRAM page misses), sequential reads + random full-line writes at ~0.7
        // (tools/micro/row_bias.hip: 4.10 vs 5.73 TB/s) -- and smaller tables gain nothing (5.58 vs 5.61).
        const int64_t spokes = h->chain_hub_item ? h->n_users : h->n_items;
        const size_t table_bytes = (size_t)spokes * (size_t)h->k * esize(h), arena_bytes = (size_t)n * (size_t)h->k * esize(h);
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        // the arena allocated at the top of the call is already OUT of free_b: count it back, or the 0.6 / 0.3 limits below would apply
        // to (free - arena) and an arena between 0.375 and 0.6 of the free memory -- north_star's 102 GB on 288 GB is within 1 % of that
        // edge -- would be allocated, judged too large, freed again and the run would silently take the table form
        if (spec.ptr && spec.bytes == arena_bytes) free_b += spec.bytes;
        const bool forced = (h->flags & CMI_FLAG_SPOKE_ARENA) || cmi_exp_env("CMI_ARENA");
        const bool large = table_bytes >= ((size_t)2 << 30) && (double)arena_bytes <= 0.6 * (double)free_b;
        // In between (spoke tables of 256 MiB .. 2 GiB: BASELINE C5's share has a 1-GiB Q) the better form depends on the BOX: the same
        // library measures the arena 9 % faster on some MI355X boxes of the pool and 4 % slower on others.  There
        // the arena is built and the first training call times one epoch of each form at learning rate 0 (which leaves every parameter
        // as it is) and keeps the faster one (arena_probe below).  CMI_ARENA_PROBE=1 forces the probe at any size (tests), =0 disables it.
        const char *pe = getenv("CMI_ARENA_PROBE");
        // (not under CMI_FLAG_STRICT unless asked for: the probe's rate-0 epochs turn a -0.0 parameter into +0.0, a bit-level drift the
        // strict path promises not to have; strict instances in that range keep the table form)
        const bool probe = !forced && !large && (pe ? atoi(pe) != 0 : (!h->strict && table_bytes >= ((size_t)256 << 20) && (double)arena_bytes <= 0.3 * (double)free_b));
        h->arena_probe = false;
        if (forced || large || probe) {
            // next_pos[p] = stream position of the next tuple of the same spoke row (its tuples sit in ascending levels, hence ascending
            // positions); the last one wraps to the first: that is where the row waits for the next epoch
            // on the device out of the uploaded stream (sched_device.hip arena_positions_device: a stable sort of the positions by spoke
            // row); the host walk (0.5 s for C3's 50 M tuples) only if that fails
            e = hipMalloc((void **)&h->d_next, (size_t)n * 4);
            if (e == hipSuccess) e = hipMalloc((void **)&h->d_first, (size_t)spokes * 4);
            if (e == hipSuccess && (hipError_t)arena_positions_device(h->stream, n, h->chain_hub_item ? h->d_su : h->d_sj, spokes, h->d_next, h->d_first) != hipSuccess) {
                (void)hipGetLastError();
                HostBuf<int32_t> nxt((size_t)n), first((size_t)spokes), back(dev_stream ? (size_t)n : 0);
                if (dev_stream) { // the stream exists on the device only: the spoke ids come back for the host walk
                    e = hipMemcpyAsync(back.data(), h->chain_hub_item ? h->d_su : h->d_sj, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
                }
                if (e == hipSuccess) { // (a failed copy-back must not be walked over, nor its error overwritten: ADVICE r5)
                    const HostBuf<int32_t> &sp = dev_stream ? back : (h->chain_hub_item ? su : sj);
                    arena_positions(n, sp.data(), spokes, nxt.data(), first.data());
                    e = hipMemcpyAsync(h->d_next, nxt.data(), (size_t)n * 4, hipMemcpyHostToDevice, h->stream);
                }
                if (e == hipSuccess) e = hipMemcpyAsync(h->d_first, first.data(), (size_t)spokes * 4, hipMemcpyHostToDevice, h->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(h->stream); // nxt / first are locals
            }
            if (e == hipSuccess) {
                h->d_arena = spec.take(arena_bytes); // allocated at the top of the call (clean memory), if the sizes had announced it
                if (!h->d_arena) e = hipMalloc(&h->d_arena, arena_bytes);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream); // nxt / first are locals
            if (e == hipSuccess) {
                h->arena_on = true;
                h->arena_valid = false;
                h->table_valid = true;
                h->arena_which = h->chain_hub_item ? CMI_STATE_P : CMI_STATE_Q;
                h->arena_probe = probe;
                h->arena_probe_table = pe && atoi(pe) < 0; // (CMI_ARENA_PROBE=-1: tests force the "table wins" verdict)
            } else if (!forced) { // not enough memory after all: run without
                (void)hipGetLastError();
                for (void **q : {(void **)&h->d_arena, (void **)&h->d_next, (void **)&h->d_first})
                    if (*q) {
                        hipFree(*q);
                        *q = nullptr;
                    }
                e = hipSuccess;
            }
        }
    }
    lap("arena");
    if (e == hipSuccess && h->owner) {
        const int32_t n_spokes = h->owner_hub_item ? h->n_users : h->n_items;
        h->own_stride = owner_record_stride(h->model, h->k, h->n_conds, h->f64, h->owner_hub_item);
        const int64_t rec_bytes = h->own_stride * 8;
        if (((int64_t)n_spokes + h->n_owners) * rec_bytes >= ((int64_t)1 << 32) - 65536) {
            free_ratings(h);
            CMI_FAIL(h, CMI_E_UNSUPPORTED, "set_ratings: owner schedule: the record table (%d records of %lld bytes) must stay below 4 GB",
                     n_spokes, (long long)rec_bytes);
        }
        // every owner's list is followed by 2 x depth inert entries (OWN_NOP) on the owner's dummy record behind the table: the kernel
        // runs whole rounds of up to `depth` steps and reads entries depth + 1 places ahead, unchecked
        const int pad = 2 * owner_depth();
        const int ncw = owner_mask_words(h->model, h->n_conds);
        const size_t rb = owner_rec_bytes(ncw);   // OwnerRecT<ncw>: {off, hub, want, flags, rating, mask[ncw]}
        std::vector<uint64_t> recs(((size_t)n + (size_t)h->n_owners * (size_t)pad) * (rb / 8), 0);
        auto put = [&](int64_t pos, uint32_t off, int32_t hub, uint32_t want, uint32_t flags, double rating, const uint64_t *mask) {
            unsigned char *q = reinterpret_cast<unsigned char *>(recs.data()) + (size_t)pos * rb;
            memcpy(q, &off, 4);
            memcpy(q + 4, &hub, 4);
            memcpy(q + 8, &want, 4);
            memcpy(q + 12, &flags, 4);
            if (h->f64) memcpy(q + 16, &rating, 8);
            else {
                const float rf = (float)rating;
                memcpy(q + 16, &rf, 4);
            }
            if (mask) memcpy(q + 24, mask, 8 * (size_t)ncw);
        };
        int64_t out = 0;
        int32_t w = 0;
        auto nop = [&](int32_t owner) {
            put(out++, (uint32_t)(((int64_t)n_spokes + owner) * rec_bytes), 0, 0, OWN_HUB_FWD | OWN_SPK_FWD | OWN_NOP, 0.0, nullptr);
        };
        for (; w < h->n_owners && osch.own_off[(size_t)w + 1] == 0; ++w) // owners without tuples ahead of the first list
            for (int i = 0; i < pad; ++i) nop(w);
        for (int64_t s = 0; s < n; ++s) {
            const int64_t t = sch.perm[(size_t)s];
            uint64_t mask[6] = {0, 0, 0, 0, 0, 0};
            if (contextual)
                for (int32_t c = ctx_ptr[ctx[t]]; c < ctx_ptr[ctx[t] + 1]; ++c) mask[ctx_conds[c] >> 6] |= (uint64_t)1 << (ctx_conds[c] & 63);
            put(out++, (uint32_t)((int64_t)(h->owner_hub_item ? u[t] : j[t]) * rec_bytes), h->owner_hub_item ? j[t] : u[t], osch.want[(size_t)s],
                osch.flags[(size_t)s], r[t], mask);
            while (w < h->n_owners && s + 1 == osch.own_off[(size_t)w + 1]) { // the end of owner w's list (and of empty owners after it)
                for (int i = 0; i < pad; ++i) nop(w);
                ++w;
            }
        }
        for (; w < h->n_owners; ++w) // owners without tuples (n == 0 never gets here)
            for (int i = 0; i < pad; ++i) nop(w);
        e = upload((void **)&h->d_own_recs, recs, h->stream);
        if (e == hipSuccess) e = upload((void **)&h->d_own_off, osch.own_off, h->stream);
        if (e == hipSuccess) e = hipMalloc(&h->d_tagged, ((size_t)n_spokes + (size_t)h->n_owners) * (size_t)h->own_stride * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&h->d_flow_err, 16);
        if (e == hipSuccess) e = hipMemsetAsync(h->d_flow_err, 0, 16, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream); // recs is a local
    }
    if (e == hipSuccess && h->model == CMI_MODEL_SVDPP) {
        // userItemsCache = train.rowColumnsCache (SVDPlusPlus.java:52): the items of every user in the 2-D train matrix, ascending
        std::vector<int32_t> ptr((size_t)h->n_users + 1, 0), items((size_t)n);
        for (int64_t t = 0; t < n; ++t) ptr[(size_t)u[t] + 1]++;
        for (int32_t x = 0; x < h->n_users; ++x) ptr[(size_t)x + 1] += ptr[(size_t)x];
        {
            std::vector<int32_t> cur(ptr.begin(), ptr.end() - 1);
            for (int64_t t = 0; t < n; ++t) items[(size_t)cur[(size_t)u[t]]++] = j[t];
        }
        for (int32_t x = 0; x < h->n_users; ++x) std::sort(items.begin() + ptr[(size_t)x], items.begin() + ptr[(size_t)x + 1]);
        e = upload((void **)&h->d_ui_ptr, ptr, h->stream);
        if (e == hipSuccess) e = upload((void **)&h->d_ui_items, items, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream); // ptr/items are locals
    }
    if (e == hipSuccess && h->n_slots > 0) {
        // zeroed: a launch shape that writes fewer slots than were reserved must not feed garbage into the loss
        e = hipMalloc((void **)&h->d_loss_part, (size_t)h->n_slots * sizeof(double));
        if (e == hipSuccess) e = hipMemsetAsync(h->d_loss_part, 0, (size_t)h->n_slots * sizeof(double), h->stream);
    }
    if (e == hipSuccess && !h->blk_off.empty()) {
        e = hipMalloc((void **)&h->d_blk_off, h->blk_off.size() * sizeof(int32_t));
        if (e == hipSuccess)
            e = hipMemcpyAsync(h->d_blk_off, h->blk_off.data(), h->blk_off.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream);
    }
    if (e == hipSuccess && h->n_tail > 0) { // the tail launches read their level offsets from the device
        e = hipMalloc((void **)&h->d_tail_off, h->level_off.size() * sizeof(int64_t));
        if (e == hipSuccess)
            e = hipMemcpyAsync(h->d_tail_off, h->level_off.data(), h->level_off.size() * sizeof(int64_t), hipMemcpyHostToDevice,
                               h->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        free_ratings(h);
        CMI_FAIL(h, CMI_E_HIP, "set_ratings: upload failed: %s", hipGetErrorString(e));
    }
    lap("rest");
    h->n = n;
    h->tuple_bytes = h->owner ? (ns + (int64_t)h->n_owners * 2 * owner_depth()) * (int64_t)owner_rec_bytes(owner_mask_words(h->model, h->n_conds))
                              : ns * (8 + (int64_t)esize(h) + 4 * (int64_t)dmax + 0) + (h->chain ? 4 * (h->n_units + 1) : 0) + (h->arena_on ? 4 * ns : 0);
    h->have_ratings = true;
    return CMI_OK;
}

extern "C" int cmi_schedule_info(cmi_handle h, int64_t info[8]) {
    if (!h || !info) return CMI_E_INVALID;
    if (!h->have_ratings) CMI_FAIL(h, CMI_E_INVALID, "schedule_info: call cmi_set_ratings first");
    info[0] = h->n_tail > 0 ? h->n_launches : h->sched_levels; // launches: a narrow run of levels shares one
    info[1] = h->max_level;
    info[2] = h->n;
    info[3] = h->dmax;
    int64_t sb = 0;
    for (int w = 0; w < CMI_STATE_COUNT; ++w) sb += h->state_count[w] * (int64_t)esize(h);
    info[4] = sb;
    info[5] = h->tuple_bytes;
    info[6] = h->owner ? (h->owner_hub_item ? 6 : 7) : (h->serial ? 1 : (h->chain ? (h->chain_hub_item ? 4 : 5) : 0));
    info[7] = h->owner ? ((int64_t)h->n_owners | ((int64_t)h->n_team << 32)) : (h->chain ? h->n_units : (h->d_blk_off ? (int64_t)h->blk_off.size() - 1 : 0));
    return CMI_OK;
}

// HBM bytes one epoch of the schedule that is actually loaded has to move, from the schedule itself (bench.py's roofline numerator since
// round 3: SURVEY 8(d)'s per-update figure assumes both rows of every tuple come from HBM, which the hub-chain kernel undercuts by
// keeping the hub row on chip along a unit).  Per UNIT of a chain schedule: the hub row, the hub's scalar bias and the hub's
// context-bias row, read and written once, plus its unit_off entry; per TUPLE: the spoke row read and written, the tuple stream
// (su, sj, sr, sconds), the spoke's scalar bias and the spoke side's context-bias cells.  Plain level schedules: both sides per tuple.
// out[0] bills every scattered scalar at the 64-byte sector it occupies on the way in and on the way out (what the memory system
// moves), out[1] at its own size (what the algorithm needs), out[2] = SURVEY 8(d)'s no-reuse figure, out[3] = 1 if hub reuse is modelled.
extern "C" int cmi_schedule_traffic(cmi_handle h, int64_t out[4]) {
    if (!h || !out) return CMI_E_INVALID;
    if (!h->have_ratings) CMI_FAIL(h, CMI_E_INVALID, "schedule_traffic: call cmi_set_ratings first");
    const int64_t e = (int64_t)esize(h), K = h->k, D = h->dmax, NC = h->n_conds, n = h->n;
    const bool bu = cmi_model_has(h->model, CMI_STATE_USER_BIAS), bj = cmi_model_has(h->model, CMI_STATE_ITEM_BIAS);
    const bool uc = cmi_model_has(h->model, CMI_STATE_UC_BIAS), ic = cmi_model_has(h->model, CMI_STATE_IC_BIAS);
    const bool cb = cmi_model_has(h->model, CMI_STATE_COND_BIAS);
    const int64_t SECT = 64;
    const int64_t row = 2 * K * e, stream = 8 + e + 4 * D;
    const int64_t row_sectors = (NC * e + SECT - 1) / SECT;              // a context-bias row, in sectors
    const int64_t cells_sect = 2 * SECT * std::min<int64_t>(D, row_sectors), cells_own = 2 * e * D; // D scattered cells of one row
    const int64_t ctx_row_sect = 2 * row_sectors * SECT, ctx_row_own = 2 * NC * e;                 // the whole row, coalesced
    const int64_t S = (bu ? 1 : 0) + (bj ? 1 : 0), T = (uc ? 1 : 0) + (ic ? 1 : 0) + (cb ? 1 : 0);
    out[2] = n * (12 + e + 4 * D + 2 * row + 2 * e * S + 2 * e * D * T);
    if (h->chain) {
        const bool hi = h->chain_hub_item;
        const bool hub_b = hi ? bj : bu, spk_b = hi ? bu : bj, hub_c = hi ? ic : uc, spk_c = hi ? uc : ic;
        const int64_t unit_sect = row + (hub_b ? 2 * SECT : 0) + (hub_c ? ctx_row_sect : 0) + 4;
        const int64_t unit_own = row + (hub_b ? 2 * e : 0) + (hub_c ? ctx_row_own : 0) + 4;
        const int64_t arena_idx = h->arena_on ? 4 : 0; // next_pos of the spoke arena
        const int64_t tup_sect = stream + arena_idx + row + (spk_b ? 2 * SECT : 0) + (spk_c ? cells_sect : 0);
        const int64_t tup_own = stream + arena_idx + row + (spk_b ? 2 * e : 0) + (spk_c ? cells_own : 0);
        out[0] = h->n_units * unit_sect + n * tup_sect;
        out[1] = h->n_units * unit_own + n * tup_own;
        out[3] = 1 | (h->arena_on ? 2 : 0);
    } else {
        out[0] = n * (stream + 2 * row + S * 2 * SECT + T * cells_sect);
        out[1] = n * (stream + 2 * row + S * 2 * e + T * cells_own);
        out[3] = 0;
    }
    return CMI_OK;
}

extern "C" const char *cmi_schedule_note(cmi_handle h) { return h ? h->sched_note.c_str() : ""; }

// ---- training -------------------------------------------------------------------------------------

template <typename T>
static SgdArgs<T> make_args(cmi_instance *h) {
    SgdArgs<T> a;
    a.P = (T *)h->state[CMI_STATE_P];
    a.Q = (T *)h->state[CMI_STATE_Q];
    a.userBias = (T *)h->state[CMI_STATE_USER_BIAS];
    a.itemBias = (T *)h->state[CMI_STATE_ITEM_BIAS];
    a.condBias = (T *)h->state[CMI_STATE_COND_BIAS];
    a.ucBias = (T *)h->state[CMI_STATE_UC_BIAS];
    a.icBias = (T *)h->state[CMI_STATE_IC_BIAS];
    a.su = h->d_su;
    a.sj = h->d_sj;
    a.sr = (const T *)h->d_sr;
    a.sconds = h->d_sconds;
    a.hp = h->d_hp;
    a.loss_part = h->d_loss_part;
    a.k = h->k;
    a.n_conds = h->n_conds;
    a.dmax = h->dmax;
    if (h->arena_on) {
        a.arena = (T *)h->d_arena;
        a.next_pos = h->d_next;
    }
#ifdef CMI_OWNER_TRACE
    a.trace = h->d_trace;
#endif
    return a;
}

#ifdef CMI_OWNER_TRACE
// Debug builds only (make TRACE=1; tools/exp/owner_trace.py): owner epochs from now on record every tuple's inputs and outputs (12 rows of
// 64 doubles per list position, owner_kernels.hip CMI_TR_ROWS); the dump writes the most recent epoch's trace to a file and stops tracing.
extern "C" int cmi_debug_owner_trace(cmi_handle h) {
    if (!h || !h->owner) return CMI_E_INVALID;
    if (hipSetDevice(h->device) != hipSuccess) return CMI_E_HIP;
    if (!h->d_trace) {
        h->trace_doubles = ((size_t)h->n + (size_t)h->n_owners * 2 * (size_t)owner_depth()) * 12 * 64;
        CMI_HIP(h, hipMalloc((void **)&h->d_trace, h->trace_doubles * 8));
    }
    CMI_HIP(h, hipMemsetAsync(h->d_trace, 0, h->trace_doubles * 8, h->stream));
    CMI_HIP(h, hipStreamSynchronize(h->stream));
    return CMI_OK;
}
extern "C" int cmi_debug_owner_trace_dump(cmi_handle h, const char *path) {
    if (!h || !h->d_trace || !path) return CMI_E_INVALID;
    if (hipSetDevice(h->device) != hipSuccess) return CMI_E_HIP;
    std::vector<double> host(h->trace_doubles);
    CMI_HIP(h, hipMemcpyAsync(host.data(), h->d_trace, h->trace_doubles * 8, hipMemcpyDeviceToHost, h->stream));
    CMI_HIP(h, hipStreamSynchronize(h->stream));
    FILE *f = fopen(path, "wb");
    if (!f) CMI_FAIL(h, CMI_E_INVALID, "cannot open %s", path);
    const size_t wrote = fwrite(host.data(), 8, host.size(), f);
    fclose(f);
    (void)hipFree(h->d_trace);
    h->d_trace = nullptr;
    return wrote == host.size() ? CMI_OK : CMI_E_INVALID;
}
#endif

// ---- spoke arena <-> model table ------------------------------------------------------------------------------------------
int cmi_sync_table_from_arena(cmi_instance *h) {
    if (!h->arena_on || h->table_valid) return CMI_OK;
    CMI_HIP(h, hipSetDevice(h->device));
    const int64_t rows = h->arena_which == CMI_STATE_P ? h->n_users : h->n_items;
    CMI_HIP(h, h->f64 ? launch_arena_gather<double>((double *)h->state[h->arena_which], (const double *)h->d_arena, h->d_first, rows, h->k, h->stream)
                      : launch_arena_gather<float>((float *)h->state[h->arena_which], (const float *)h->d_arena, h->d_first, rows, h->k, h->stream));
    h->table_valid = true;
    return CMI_OK;
}
static int sync_arena_from_table(cmi_instance *h) {
    if (!h->arena_on || h->arena_valid) return CMI_OK;
    const int64_t rows = h->arena_which == CMI_STATE_P ? h->n_users : h->n_items;
    CMI_HIP(h, h->f64 ? launch_arena_scatter<double>((const double *)h->state[h->arena_which], (double *)h->d_arena, h->d_first, rows, h->k, h->stream)
                      : launch_arena_scatter<float>((const float *)h->state[h->arena_which], (float *)h->d_arena, h->d_first, rows, h->k, h->stream));
    h->arena_valid = true;
    return CMI_OK;
}

template <typename T>
static ExtArgs<T> make_ext_args(cmi_instance *h) {
    ExtArgs<T> a;
    a.P = (T *)h->state[CMI_STATE_P];
    a.Q = (T *)h->state[CMI_STATE_Q];
    a.userBias = (T *)h->state[CMI_STATE_USER_BIAS];
    a.itemBias = (T *)h->state[CMI_STATE_ITEM_BIAS];
    a.Y = (T *)h->state[CMI_STATE_Y];
    a.cc = (T *)h->state[CMI_STATE_CC_MATRIX];
    a.cf = (T *)h->state[CMI_STATE_CF_MATRIX];
    a.cv = (T *)h->state[CMI_STATE_C_VECTOR];
    a.su = h->d_su;
    a.sj = h->d_sj;
    a.sr = (const T *)h->d_sr;
    a.sconds = h->d_sconds;
    a.empty_conds = h->d_empty;
    a.ui_ptr = h->d_ui_ptr;
    a.ui_items = h->d_ui_items;
    a.hp = h->d_hp;
    a.upbound = 1.0 / std::sqrt((double)h->n_ctx_dims); // CAMF_MCS.java:47
    a.lowbound = 1.0 / std::pow(10.0, 100.0);             // CAMF_MCS.java:48
    a.k = h->k;
    a.n_conds = h->n_conds;
    a.dmax = h->dmax;
    a.num_f = h->num_f;
    a.n_empty = (int32_t)h->empty_conds.size();
    return a;
}

// enqueue every level of one epoch + the loss reduction on h->stream
static hipError_t enqueue_levels(cmi_instance *h) {
    LaunchCfg cfg{h->model, h->strict};
    hipError_t e = hipSuccess;
    const int64_t n_levels = (int64_t)h->level_off.size() - 1;
    if (is_ext_model(h->model)) {
        if (h->f64) return launch_ext_serial<double>(make_ext_args<double>(h), h->model, h->strict, h->n, h->d_loss, h->stream);
        return launch_ext_serial<float>(make_ext_args<float>(h), h->model, h->strict, h->n, h->d_loss, h->stream);
    }
    if (h->serial) {
        if (h->d_blk_off) { // CAMF_C over conflict-free blocks
            const int nb = (int)h->blk_off.size() - 1;
            if (h->f64) return launch_camfc_blocks<double>(make_args<double>(h), h->d_blk_off, nb, h->d_loss, h->stream);
            return launch_camfc_blocks<float>(make_args<float>(h), h->d_blk_off, nb, h->d_loss, h->stream);
        }
        if (h->f64) return launch_serial<double>(make_args<double>(h), cfg, h->n, h->d_loss, h->stream);
        return launch_serial<float>(make_args<float>(h), cfg, h->n, h->d_loss, h->stream);
    }
    if (h->owner) {
        // The owner epoch is a persistent launch whose workgroups wait for each other: every one of them has to be resident.  Two of
        // them in flight at once (two folds of `cv -p on` on one GPU, each on its own stream) could each hold part of the compute units
        // and wait forever for the rest, so the gate (OwnerDeviceLock) admits owner epochs of this process only while their workgroups
        // fit the device together, and keeps another PROCESS's epochs away through an advisory file lock; it is held from the launch
        // until the stream has drained (cmi_train_epoch_async is synchronous for this schedule).  Should a foreign persistent kernel
        // hold compute units anyway, the waits are bounded, the first one to expire ends every other wait early (owner_spin_expired)
        // and the epoch is reported as failed right here -- also on the cmi_train_epoch_async path, which never calls cmi_last_loss.
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || cus < 1) cus = 1;
        OwnerDeviceLock guard(h->device, h->n_team + (h->n_owners - h->n_team + 3) / 4, cus);
        if (!guard.ok) {
            h->owner_busy = true;
            h->owner_busy_why = guard.why;
            return hipErrorNotReady;
        }
        const int32_t n_spokes = h->owner_hub_item ? h->n_users : h->n_items;
        // the epoch's tag base: an odd multiple of the sequence number mod 2^32 -- two epochs' tags of one row differ by far more than a
        // row has updates, so a record copy from an earlier epoch cannot carry the tag a tuple of this epoch waits for
        const uint32_t tag0 = (uint32_t)(++h->owner_epoch_seq) * 0x9E3779B1u;
        e = h->f64 ? launch_owner_epoch<double>(make_args<double>(h), h->model, h->owner_hub_item, h->strict, h->d_own_recs, h->d_own_off, h->n_owners, h->n_team, h->d_tagged,
                                                h->own_stride, n_spokes, h->d_flow_err, tag0, h->stream)
                   : launch_owner_epoch<float>(make_args<float>(h), h->model, h->owner_hub_item, false, h->d_own_recs, h->d_own_off, h->n_owners, h->n_team, h->d_tagged,
                                               h->own_stride, n_spokes, h->d_flow_err, tag0, h->stream);
        if (e == hipSuccess) e = launch_reduce_loss(h->d_loss_part, h->n_slots, h->d_scratch, h->d_loss, h->stream);
        int32_t stalled = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&stalled, h->d_flow_err, 4, hipMemcpyDeviceToHost, h->stream); // (the instance's stream, not the legacy one)
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) {
            if (stalled) {
                h->owner_stalled = true;
                e = hipErrorLaunchFailure;
            }
        }
        return e;
    }
    if (h->chain) {
        for (int64_t l = 0; l < n_levels && e == hipSuccess; ++l) {
            const int64_t b = h->level_off[(size_t)l];
            const int cnt = (int)(h->level_off[(size_t)l + 1] - b);
            e = h->f64 ? launch_chain_level<double>(make_args<double>(h), cfg, h->chain_hub_item, h->d_unit_off, b, cnt, h->slot_off[(size_t)l], h->stream)
                       : launch_chain_level<float>(make_args<float>(h), cfg, h->chain_hub_item, h->d_unit_off, b, cnt, h->slot_off[(size_t)l], h->stream);
        }
    } else if (h->f64) {
        const SgdArgs<double> a = make_args<double>(h);
        for (int64_t l = 0; l < n_levels && e == hipSuccess; ++l) {
            const int32_t run = h->tail_len.empty() ? 0 : h->tail_len[(size_t)l];
            if (run > 0) {
                e = launch_tail<double>(a, cfg, h->d_tail_off + l, run, h->slot_off[(size_t)l], h->stream);
                l += run - 1;
                continue;
            }
            e = launch_level_generic<double>(a, cfg, h->level_off[(size_t)l],
                                             (int)(h->level_off[(size_t)l + 1] - h->level_off[(size_t)l]),
                                             h->slot_off[(size_t)l], h->stream);
        }
    } else {
        const SgdArgs<float> a = make_args<float>(h);
        for (int64_t l = 0; l < n_levels && e == hipSuccess; ++l) {
            const int32_t run = h->tail_len.empty() ? 0 : h->tail_len[(size_t)l];
            if (run > 0) {
                e = launch_tail_f32(a, cfg, h->fast ? 1 : (h->small ? 2 : 0), h->d_tail_off + l, run, h->slot_off[(size_t)l], h->stream);
                l += run - 1;
                continue;
            }
            const int64_t b = h->level_off[(size_t)l];
            const int cnt = (int)(h->level_off[(size_t)l + 1] - b);
            e = h->fast    ? launch_level_fast_f32(a, cfg, b, cnt, h->slot_off[(size_t)l], h->stream)
                : h->small ? launch_level_small_f32(a, cfg, b, cnt, h->slot_off[(size_t)l], h->stream)
                           : launch_level_generic<float>(a, cfg, b, cnt, h->slot_off[(size_t)l], h->stream);
        }
    }
    if (e == hipSuccess) e = launch_reduce_loss(h->d_loss_part, h->n_slots, h->d_scratch, h->d_loss, h->stream);
    return e;
}

static int arena_probe(cmi_instance *h);

static int enqueue_epoch(cmi_instance *h, double lrate, bool probing = false) {
    if (!h->have_ratings) CMI_FAIL(h, CMI_E_INVALID, "train: call cmi_set_ratings first");
    CMI_HIP(h, hipSetDevice(h->device));
    if (h->arena_probe && !probing)
        if (int rc = arena_probe(h)) return rc;
    h->hp.lr = lrate;
    CMI_HIP(h, launch_set_hparams(h->d_hp, h->hp, h->stream));
    if (h->n == 0) {
        CMI_HIP(h, hipMemsetAsync(h->d_loss, 0, sizeof(double), h->stream));
        h->epoch_timed = false;
        return CMI_OK;
    }
    // a graph of several hundred thousand kernel nodes is neither instantiable in reasonable time nor useful
    const bool graph = h->use_graph && !h->serial && !h->owner && (h->n_tail > 0 ? h->n_launches : (int64_t)h->level_off.size() - 1) <= 65536;
    if (graph && !h->graph_exec) {
        hipGraph_t g = nullptr;
        CMI_HIP(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        hipError_t e = enqueue_levels(h);
        hipError_t e2 = hipStreamEndCapture(h->stream, &g);
        if (e != hipSuccess || e2 != hipSuccess) {
            if (g) hipGraphDestroy(g);
            CMI_FAIL(h, CMI_E_HIP, "graph capture failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        }
        e = hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (e != hipSuccess) {
            h->graph_exec = nullptr;
            CMI_FAIL(h, CMI_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
        }
    }
    if (h->arena_on) {
        if (int rc = sync_arena_from_table(h)) return rc; // (outside the captured graph: only after the table was rewritten)
        h->table_valid = false;                          // this epoch moves the spoke rows inside the arena only
    }
    CMI_HIP(h, hipEventRecord(h->ev0, h->stream));
    if (graph) CMI_HIP(h, hipGraphLaunch(h->graph_exec, h->stream));
    else {
        const hipError_t e = enqueue_levels(h);
        if (h->owner_busy) {
            h->owner_busy = false;
            CMI_FAIL(h, CMI_E_BUSY, "owner epoch not started on device %d: %s (persistent kernels of two processes must not share a device); "
                     "the model is untouched -- retry, or use CMI_FLAG_NO_OWNER", h->device, h->owner_busy_why);
        }
        if (h->owner_stalled)
            CMI_FAIL(h, CMI_E_HIP, "owner epoch stalled: a tuple waited past its bound for a predecessor's record (is another persistent kernel "
                     "holding compute units of device %d?) -- the model state is invalid; reload it and use CMI_FLAG_NO_OWNER", h->device);
        CMI_HIP(h, e);
    }
    CMI_HIP(h, hipEventRecord(h->ev1, h->stream));
    h->epoch_timed = true;
    return CMI_OK;
}

// Table or arena for the spoke rows of this box (see cmi_set_ratings): one timed epoch of each form at learning rate 0 -- every update
// is then x + 0 * (...) = x, so the model is what it was (to the sign of an exact zero) -- after one untimed epoch that also builds the
// form's graph.  The arena stays if it is at least 3 % faster; otherwise its memory is released.
static int arena_probe(cmi_instance *h) {
    h->arena_probe = false;
    float ms[2] = {0.f, 0.f};
    for (int form = 0; form < 2; ++form) {
        h->arena_on = form == 1;
        if (h->graph_exec) {
            hipGraphExecDestroy(h->graph_exec);
            h->graph_exec = nullptr;
        }
        for (int rep = 0; rep < 2; ++rep)
            if (int rc = enqueue_epoch(h, 0.0, true)) return rc;
        CMI_HIP(h, hipStreamSynchronize(h->stream));
        CMI_HIP(h, hipEventElapsedTime(&ms[form], h->ev0, h->ev1));
        if (form == 1)
            if (int rc = cmi_sync_table_from_arena(h)) return rc; // the tables are the masters again, whichever form wins
    }
    const bool keep = ms[1] < 0.97f * ms[0] && !h->arena_probe_table;
    char note[200];
    snprintf(note, sizeof note, "spoke arena probe: table %.2f ms, arena %.2f ms per epoch -> %s", ms[0], ms[1], keep ? "arena" : "table");
    h->sched_note = h->sched_note.empty() ? note : h->sched_note + "; " + note;
    if (!keep) {
        if (h->graph_exec) { // captured in the arena form (null when the epoch is not graph-captured: CMI_FLAG_NO_GRAPH, > 65536 launches)
            hipGraphExecDestroy(h->graph_exec);
            h->graph_exec = nullptr;
        }
        h->arena_on = h->arena_valid = false;
        h->table_valid = true;
        for (void **q : {(void **)&h->d_arena, (void **)&h->d_next, (void **)&h->d_first})
            if (*q) {
                hipFree(*q);
                *q = nullptr;
            }
    }
    return CMI_OK;
}

extern "C" int cmi_train_epoch_async(cmi_handle h, double lrate) {
    if (!h) return CMI_E_INVALID;
    return enqueue_epoch(h, lrate);
}

extern "C" int cmi_last_loss(cmi_handle h, double *loss_out) {
    if (!h || !loss_out) return CMI_E_INVALID;
    CMI_HIP(h, hipSetDevice(h->device));
    CMI_HIP(h, hipMemcpyAsync(h->h_loss, h->d_loss, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    int32_t flow_stat[4] = {0, 0, 0, 0};
    if (h->owner) CMI_HIP(h, hipMemcpyAsync(flow_stat, h->d_flow_err, 16, hipMemcpyDeviceToHost, h->stream));
    CMI_HIP(h, hipStreamSynchronize(h->stream));
    const int32_t flow_err = flow_stat[0];
    if (h->owner && h->n_team > 0 && getenv("CMI_OWNER_STATS"))
        fprintf(stderr, "[cmi] owner teams: %d; cumulative, busiest owner: compute wave found the ring empty %d times, loader found it full %d times\n",
                h->n_team, flow_stat[1], flow_stat[2]);
    else if (h->owner && getenv("CMI_OWNER_STATS"))
        fprintf(stderr, "[cmi] owner: cumulative -- busiest owner (%lld tuples per epoch) found %d records not ready and polled %d times; all owners: %d "
                "records not ready (of %lld tuples per epoch)\n", (long long)h->max_level, flow_stat[1], flow_stat[2], flow_stat[3], (long long)h->n);
    if (flow_err) CMI_FAIL(h, CMI_E_HIP, "owner epoch stalled: a tuple waited past its bound for a predecessor (model state is invalid)");
    h->last_loss = *h->h_loss;
    *loss_out = h->last_loss;
    return CMI_OK;
}

extern "C" int cmi_train_epoch(cmi_handle h, double lrate, double *loss_out) {
    if (!h) return CMI_E_INVALID;
    if (int rc = enqueue_epoch(h, lrate)) return rc;
    double loss = 0.0;
    if (int rc = cmi_last_loss(h, &loss)) return rc;
    if (loss_out) *loss_out = loss;
    return CMI_OK;
}

// IterativeRecommender.isConverged + updateLRate (IterativeRecommender.java:145-229), host side.  first_iter / last_loss let a
// reloaded model (cmi_load_model) continue the loop exactly where it stopped: the bold driver compares with the previous epoch's
// loss and is off in iteration 1 only.
// (shared with cmi_group_train_from: `epoch` runs one epoch at the given rate and returns the loss the host steers by)
int cmi_train_loop(const std::function<int(double, double *)> &epoch, std::string &err, int first_iter, double prev_loss, int num_iters,
                   double init_lrate, double max_lrate, int bold_driver, double decay, int early_stop, double *losses, double *lrates,
                   int *iters_run, double *final_lrate) {
    auto fail = [&](int code, const char *msg) {
        err = msg;
        return code;
    };
    if (first_iter < 1) return fail(CMI_E_INVALID, "train: first_iter must be >= 1");
    if (early_stop != 0 && early_stop != 1) return fail(CMI_E_UNSUPPORTED, "train: early_stop must be 0 (none) or 1 (loss)");
    double lr = init_lrate, last_loss = prev_loss, measure = 0.0, last_measure = 0.0;
    if (early_stop == 1) last_measure = measure = prev_loss;
    if (iters_run) *iters_run = 0;
    for (int n = 0; n < num_iters; ++n) {
        const int it = first_iter + n;
        double loss = 0.0;
        if (lrates) lrates[n] = lr;
        if (int rc = epoch(lr, &loss)) return rc;
        if (losses) losses[n] = loss;
        if (iters_run) *iters_run = n + 1;
        if (early_stop == 1) {
            measure = loss;
            last_measure = last_loss;
        }
        const float delta_measure = (float)(last_measure - measure);
        if (std::isnan(loss) || std::isinf(loss)) {
            if (final_lrate) *final_lrate = lr;
            char buf[160];
            snprintf(buf, sizeof buf, "Loss = NaN or Infinity at iteration %d: current settings do not fit the recommender", it);
            return fail(CMI_E_NUMERIC, buf);
        }
        const bool converged = std::fabs(loss) < 1e-5 || (delta_measure > 0 && delta_measure < 1e-5);
        if (!converged && lr > 0) {
            if (bold_driver && it > 1) lr = std::fabs(last_loss) > std::fabs(loss) ? lr * 1.05 : lr * 0.5;
            else if (decay > 0 && decay < 1) lr *= decay;
            if (max_lrate > 0 && lr > max_lrate) lr = max_lrate;
        }
        last_loss = loss;
        last_measure = measure;
        if (converged) break;
    }
    if (final_lrate) *final_lrate = lr;
    return CMI_OK;
}

extern "C" int cmi_train_from(cmi_handle h, int first_iter, double prev_loss, int num_iters, double init_lrate, double max_lrate,
                              int bold_driver, double decay, int early_stop, double *losses, double *lrates, int *iters_run,
                              double *final_lrate) {
    if (!h) return CMI_E_INVALID;
    std::string err;
    const int rc = cmi_train_loop([h](double lr, double *loss) { return cmi_train_epoch(h, lr, loss); }, err, first_iter, prev_loss,
                                  num_iters, init_lrate, max_lrate, bold_driver, decay, early_stop, losses, lrates, iters_run, final_lrate);
    if (rc != CMI_OK && !err.empty()) h->err = err;
    return rc;
}

extern "C" int cmi_train(cmi_handle h, int num_iters, double init_lrate, double max_lrate, int bold_driver,
                         double decay, int early_stop, double *losses, double *lrates, int *iters_run,
                         double *final_lrate) {
    return cmi_train_from(h, 1, 0.0, num_iters, init_lrate, max_lrate, bold_driver, decay, early_stop, losses, lrates, iters_run, final_lrate);
}

extern "C" int cmi_stream(cmi_handle h, void **stream) {
    if (!h || !stream) return CMI_E_INVALID;
    *stream = (void *)h->stream;
    return CMI_OK;
}

// ---- multi-GPU exchange plumbing ----------------------------------------------------------------------
extern "C" int cmi_exchange_setup(cmi_handle h, int64_t pad_to, void **bucket, int64_t *count) {
    if (!h || !bucket || !count || pad_to < 1) return CMI_E_INVALID;
    if (h->model == CMI_MODEL_CAMF_C) CMI_FAIL(h, CMI_E_UNSUPPORTED, "exchange: CAMF_C shares condBias between all tuples and is not sharded");
    // SVD++ (Y) and CAMF_ICS / LCS / MCS (ccMatrix, cfMatrix, cVector) update containers every tuple reads: they are not in the bucket,
    // so a merge would silently leave them diverged between the ranks
    if (is_ext_model(h->model)) CMI_FAIL(h, CMI_E_UNSUPPORTED, "exchange: model %d is a single serial chain and is not sharded", h->model);
    CMI_HIP(h, hipSetDevice(h->device));
    if (int rc = cmi_sync_table_from_arena(h)) return rc;
    CMI_HIP(h, hipStreamSynchronize(h->stream));
    if (h->d_xbucket) hipFree(h->d_xbucket);
    if (h->d_xsnap) hipFree(h->d_xsnap);
    h->d_xbucket = h->d_xsnap = nullptr;
    h->x_which.clear();
    h->x_off.clear();
    int64_t off = 0;
    for (int w : {CMI_STATE_Q, CMI_STATE_ITEM_BIAS, CMI_STATE_IC_BIAS}) {
        if (!cmi_model_has(h->model, w)) continue;
        h->x_which.push_back(w);
        h->x_off.push_back(off);
        off += (h->state_count[w] + 3) / 4 * 4;
    }
    {   // total: a multiple of pad_to (even split over the ranks of a reduce-scatter) and of 4 elements (16-byte vectors)
        int64_t g = pad_to, r4 = 4;
        while (r4) {
            const int64_t t = g % r4;
            g = r4;
            r4 = t;
        }
        const int64_t step = pad_to / g * 4;
        off = (off + step - 1) / step * step;
    }
    h->x_count = off;
    const size_t bytes = (size_t)off * esize(h);
    CMI_HIP(h, hipMalloc(&h->d_xbucket, bytes));
    CMI_HIP(h, hipMalloc(&h->d_xsnap, bytes));
    CMI_HIP(h, hipMemsetAsync(h->d_xbucket, 0, bytes, h->stream));
    CMI_HIP(h, hipMemsetAsync(h->d_xsnap, 0, bytes, h->stream));
    for (size_t i = 0; i < h->x_which.size(); ++i) {
        const int w = h->x_which[i];
        CMI_HIP(h, hipMemcpyAsync((char *)h->d_xsnap + (size_t)h->x_off[i] * esize(h), h->state[w], (size_t)h->state_count[w] * esize(h),
                                  hipMemcpyDeviceToDevice, h->stream));
    }
    CMI_HIP(h, hipStreamSynchronize(h->stream));
    *bucket = h->d_xbucket;
    *count = h->x_count;
    return CMI_OK;
}

extern "C" int cmi_exchange_pack(cmi_handle h) {
    if (!h) return CMI_E_INVALID;
    if (!h->d_xbucket) CMI_FAIL(h, CMI_E_INVALID, "exchange_pack: call cmi_exchange_setup first");
    CMI_HIP(h, hipSetDevice(h->device));
    if (h->arena_on && h->arena_which == CMI_STATE_Q) // the item side is the spoke side (hub = user): Q's live rows are in the arena
        if (int rc = cmi_sync_table_from_arena(h)) return rc;
    for (size_t i = 0; i < h->x_which.size(); ++i) {
        const int w = h->x_which[i];
        const size_t ob = (size_t)h->x_off[i] * esize(h);
        CMI_HIP(h, launch_delta_pack(h->state[w], (char *)h->d_xsnap + ob, (char *)h->d_xbucket + ob, h->state_count[w], h->f64, h->stream));
    }
    return CMI_OK;
}

extern "C" int cmi_exchange_apply(cmi_handle h, double scale) {
    if (!h) return CMI_E_INVALID;
    if (!h->d_xbucket) CMI_FAIL(h, CMI_E_INVALID, "exchange_apply: call cmi_exchange_setup first");
    CMI_HIP(h, hipSetDevice(h->device));
    if (h->arena_on && h->arena_which == CMI_STATE_Q) { // Q is rewritten below: the arena's copy is stale until the next epoch re-scatters
        if (int rc = cmi_sync_table_from_arena(h)) return rc;
        h->arena_valid = false;
    }
    for (size_t i = 0; i < h->x_which.size(); ++i) {
        const int w = h->x_which[i];
        const size_t ob = (size_t)h->x_off[i] * esize(h);
        CMI_HIP(h, launch_delta_apply(h->state[w], (char *)h->d_xsnap + ob, (char *)h->d_xbucket + ob, scale, h->state_count[w], h->f64, h->stream));
    }
    return CMI_OK;
}

extern "C" int cmi_loss_device_ptr(cmi_handle h, void **ptr) {
    if (!h || !ptr) return CMI_E_INVALID;
    *ptr = h->d_loss;
    return CMI_OK;
}

extern "C" int cmi_synchronize(cmi_handle h) {
    if (!h) return CMI_E_INVALID;
    CMI_HIP(h, hipSetDevice(h->device));
    CMI_HIP(h, hipStreamSynchronize(h->stream));
    return CMI_OK;
}

extern "C" int cmi_last_epoch_ms(cmi_handle h, float *ms) {
    if (!h || !ms) return CMI_E_INVALID;
    if (!h->epoch_timed) CMI_FAIL(h, CMI_E_INVALID, "last_epoch_ms: no epoch has run");
    CMI_HIP(h, hipSetDevice(h->device));
    CMI_HIP(h, hipEventSynchronize(h->ev1));
    CMI_HIP(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return CMI_OK;
}

// ---- predict / evalRatings -------------------------------------------------------------------------

template <typename T>
cmi::ExtEvalArgs<T> cmi_ext_eval_args(cmi_instance *h, const int32_t *du, const int32_t *dj, const int32_t *dctx, const double *dr,
                                      double *dpreds, double *dpart, int bound, double lo, double hi, double min_rate) {
    ExtEvalArgs<T> x;
    x.P = (const T *)h->state[CMI_STATE_P];
    x.Q = (const T *)h->state[CMI_STATE_Q];
    x.userBias = (const T *)h->state[CMI_STATE_USER_BIAS];
    x.itemBias = (const T *)h->state[CMI_STATE_ITEM_BIAS];
    x.Y = (const T *)h->state[CMI_STATE_Y];
    x.cc = (const T *)h->state[CMI_STATE_CC_MATRIX];
    x.cf = (const T *)h->state[CMI_STATE_CF_MATRIX];
    x.cv = (const T *)h->state[CMI_STATE_C_VECTOR];
    x.u = du;
    x.j = dj;
    x.ctx = dctx;
    x.r = dr;
    x.ctx_ptr = h->d_ctx_ptr;
    x.ctx_conds = h->d_ctx_conds;
    x.empty_conds = h->d_empty;
    x.ui_ptr = h->d_ui_ptr;
    x.ui_items = h->d_ui_items;
    x.preds = dpreds;
    x.part = dpart;
    x.gm = h->hp.gm;
    x.lo = lo;
    x.hi = hi;
    x.min_rate = min_rate;
    x.k = h->k;
    x.n_conds = h->n_conds;
    x.num_f = h->num_f;
    x.n_empty = (int32_t)h->empty_conds.size();
    x.bound = bound;
    x.model = h->model;
    return x;
}
template cmi::ExtEvalArgs<float> cmi_ext_eval_args<float>(cmi_instance *, const int32_t *, const int32_t *, const int32_t *, const double *, double *, double *, int, double, double, double);
template cmi::ExtEvalArgs<double> cmi_ext_eval_args<double>(cmi_instance *, const int32_t *, const int32_t *, const int32_t *, const double *, double *, double *, int, double, double, double);

template <typename T>
static hipError_t run_eval(cmi_instance *h, int64_t n, const int32_t *du, const int32_t *dj, const int32_t *dctx,
                           const double *dr, double *dpreds, double *dpart, int bound, double lo, double hi,
                           double min_rate) {
    if (is_ext_model(h->model))
        return launch_ext_eval<T>(cmi_ext_eval_args<T>(h, du, dj, dctx, dr, dpreds, dpart, bound, lo, hi, min_rate), n, h->stream);
    EvalArgs<T> a;
    a.P = (const T *)h->state[CMI_STATE_P];
    a.Q = (const T *)h->state[CMI_STATE_Q];
    a.userBias = (const T *)h->state[CMI_STATE_USER_BIAS];
    a.itemBias = (const T *)h->state[CMI_STATE_ITEM_BIAS];
    a.condBias = (const T *)h->state[CMI_STATE_COND_BIAS];
    a.ucBias = (const T *)h->state[CMI_STATE_UC_BIAS];
    a.icBias = (const T *)h->state[CMI_STATE_IC_BIAS];
    a.u = du;
    a.j = dj;
    a.ctx = dctx;
    a.r = dr;
    a.ctx_ptr = h->d_ctx_ptr;
    a.ctx_conds = h->d_ctx_conds;
    a.preds = dpreds;
    a.part = dpart;
    a.gm = h->hp.gm;
    a.lo = lo;
    a.hi = hi;
    a.min_rate = min_rate;
    a.k = h->k;
    a.n_conds = h->n_conds;
    a.bound = bound;
    a.model = h->model;
    return launch_eval<T>(a, n, h->stream);
}

static int eval_common(cmi_instance *h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx,
                       const double *r, int bound, double lo, double hi, double min_rate, double *preds_out,
                       double sums[5]) {
    const bool contextual = !is_2d_model(h->model);
    if (h->model == CMI_MODEL_SVDPP && !h->have_ratings) CMI_FAIL(h, CMI_E_INVALID, "eval: SVD++ predicts with the users' training items; call cmi_set_ratings first");
    if (n < 0 || (n > 0 && (!u || !j))) CMI_FAIL(h, CMI_E_INVALID, "eval: null tuple arrays");
    if (contextual && n > 0 && !ctx) CMI_FAIL(h, CMI_E_INVALID, "eval: ctx required for model %d", h->model);
    if (contextual && !h->have_ratings) CMI_FAIL(h, CMI_E_INVALID, "eval: the context table comes from cmi_set_ratings; call it first");
    for (int64_t t = 0; t < n; ++t) {
        if (u[t] < 0 || u[t] >= h->n_users || j[t] < 0 || j[t] >= h->n_items)
            CMI_FAIL(h, CMI_E_INVALID, "eval: user/item id out of range at tuple %lld", (long long)t);
        if (contextual && (ctx[t] < 0 || ctx[t] >= h->n_ctx))
            CMI_FAIL(h, CMI_E_INVALID, "eval: context id %d out of range at tuple %lld", ctx[t], (long long)t);
    }
    for (int c = 0; c < 5; ++c) sums[c] = 0.0;
    if (n == 0) return CMI_OK;
    CMI_HIP(h, hipSetDevice(h->device));
    if (int rc = cmi_sync_table_from_arena(h)) return rc;
    int32_t *du = nullptr, *dj = nullptr, *dctx = nullptr;
    double *dr = nullptr, *dpreds = nullptr, *dpart = nullptr;
    const int blocks = eval_blocks(n);
    std::vector<double> part((size_t)blocks * 5);
    hipError_t e = hipMalloc((void **)&du, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&dj, (size_t)n * 4);
    if (e == hipSuccess && contextual) e = hipMalloc((void **)&dctx, (size_t)n * 4);
    if (e == hipSuccess && r) e = hipMalloc((void **)&dr, (size_t)n * 8);
    if (e == hipSuccess && preds_out) e = hipMalloc((void **)&dpreds, (size_t)n * 8);
    if (e == hipSuccess && r) e = hipMalloc((void **)&dpart, part.size() * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(du, u, (size_t)n * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dj, j, (size_t)n * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && contextual) e = hipMemcpyAsync(dctx, ctx, (size_t)n * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && r) e = hipMemcpyAsync(dr, r, (size_t)n * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess)
        e = h->f64 ? run_eval<double>(h, n, du, dj, dctx, dr, dpreds, dpart, bound, lo, hi, min_rate)
                   : run_eval<float>(h, n, du, dj, dctx, dr, dpreds, dpart, bound, lo, hi, min_rate);
    if (e == hipSuccess && preds_out) e = hipMemcpyAsync(preds_out, dpreds, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && r) e = hipMemcpyAsync(part.data(), dpart, part.size() * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    void *ptrs[] = {du, dj, dctx, dr, dpreds, dpart};
    for (void *p : ptrs)
        if (p) hipFree(p);
    CMI_HIP(h, e);
    if (r)
        for (int b = 0; b < blocks; ++b)
            for (int c = 0; c < 5; ++c) sums[c] += part[(size_t)b * 5 + c];
    return CMI_OK;
}

extern "C" int cmi_predict_batch(cmi_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx,
                                 int bound, double lo, double hi, double *out) {
    if (!h) return CMI_E_INVALID;
    if (n > 0 && !out) CMI_FAIL(h, CMI_E_INVALID, "predict_batch: null output");
    double sums[5];
    return eval_common(h, n, u, j, ctx, nullptr, bound, lo, hi, 1.0, out, sums);
}

extern "C" int cmi_eval_ratings(cmi_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx,
                                const double *r, double min_rate, double max_rate, double *out, int64_t *count) {
    if (!h) return CMI_E_INVALID;
    if (!out || (n > 0 && !r)) CMI_FAIL(h, CMI_E_INVALID, "eval_ratings: null argument");
    double sums[5];
    if (int rc = eval_common(h, n, u, j, ctx, r, 1, min_rate, max_rate, min_rate, nullptr, sums)) return rc;
    const double cnt = sums[4];
    const double mae = sums[0] / cnt;
    out[0] = mae;
    out[1] = std::sqrt(sums[1] / cnt);
    out[2] = mae / (max_rate - min_rate);
    out[3] = sums[2] / cnt;
    out[4] = std::sqrt(sums[3] / cnt);
    if (count) *count = (int64_t)cnt;
    return CMI_OK;
}

// the five sums behind cmi_eval_ratings (sum|e|, sum e^2, rounded forms, count), so that cmi_group_eval_ratings can merge shards exactly
int cmi_eval_sums(cmi_instance *h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, const double *r, double min_rate,
                  double max_rate, double sums[5]) {
    return eval_common(h, n, u, j, ctx, r, 1, min_rate, max_rate, min_rate, nullptr, sums);
}

// ---- host-only schedule export ------------------------------------------------------------------------

// ---- resident test tuples: `--early-stop MAE|RMSE` evaluates the test set after EVERY epoch (IterativeRecommender.java:
// 156-161); uploading it once instead of per call keeps that loop on the device

extern "C" int cmi_set_eval_ratings(cmi_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx,
                                    const double *r) {
    if (!h) return CMI_E_INVALID;
    const bool contextual = !is_2d_model(h->model);
    if (n < 0 || (n > 0 && (!u || !j || !r || (contextual && !ctx)))) CMI_FAIL(h, CMI_E_INVALID, "set_eval_ratings: null arrays");
    if (contextual && !h->have_ratings) CMI_FAIL(h, CMI_E_INVALID, "set_eval_ratings: the context table comes from cmi_set_ratings; call it first");
    for (int64_t t = 0; t < n; ++t) {
        if (u[t] < 0 || u[t] >= h->n_users || j[t] < 0 || j[t] >= h->n_items)
            CMI_FAIL(h, CMI_E_INVALID, "set_eval_ratings: user/item id out of range at tuple %lld", (long long)t);
        if (contextual && (ctx[t] < 0 || ctx[t] >= h->n_ctx))
            CMI_FAIL(h, CMI_E_INVALID, "set_eval_ratings: context id %d out of range at tuple %lld", ctx[t], (long long)t);
    }
    CMI_HIP(h, hipSetDevice(h->device));
    CMI_HIP(h, hipStreamSynchronize(h->stream));
    free_eval_set(h);
    if (n == 0) return CMI_OK;
    hipError_t e = hipMalloc((void **)&h->d_eu, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_ej, (size_t)n * 4);
    if (e == hipSuccess && contextual) e = hipMalloc((void **)&h->d_ectx, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_er, (size_t)n * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_epart, (size_t)eval_blocks(n) * 5 * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(h->d_eu, u, (size_t)n * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h->d_ej, j, (size_t)n * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && contextual) e = hipMemcpyAsync(h->d_ectx, ctx, (size_t)n * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h->d_er, r, (size_t)n * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        free_eval_set(h);
        CMI_FAIL(h, CMI_E_HIP, "set_eval_ratings: %s", hipGetErrorString(e));
    }
    h->n_eval = n;
    return CMI_OK;
}

int cmi_eval_resident_sums(cmi_instance *h, double min_rate, double max_rate, double sums[5]) {
    if (h->n_eval <= 0) CMI_FAIL(h, CMI_E_INVALID, "eval_resident: call cmi_set_eval_ratings first");
    CMI_HIP(h, hipSetDevice(h->device));
    if (int rc = cmi_sync_table_from_arena(h)) return rc;
    const int64_t n = h->n_eval;
    const int blocks = eval_blocks(n);
    std::vector<double> part((size_t)blocks * 5);
    hipError_t e = h->f64 ? run_eval<double>(h, n, h->d_eu, h->d_ej, h->d_ectx, h->d_er, nullptr, h->d_epart, 1, min_rate, max_rate, min_rate)
                          : run_eval<float>(h, n, h->d_eu, h->d_ej, h->d_ectx, h->d_er, nullptr, h->d_epart, 1, min_rate, max_rate, min_rate);
    if (e == hipSuccess) e = hipMemcpyAsync(part.data(), h->d_epart, part.size() * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    CMI_HIP(h, e);
    for (int c = 0; c < 5; ++c) sums[c] = 0.0;
    for (int b = 0; b < blocks; ++b)
        for (int c = 0; c < 5; ++c) sums[c] += part[(size_t)b * 5 + c];
    return CMI_OK;
}

extern "C" int cmi_eval_resident(cmi_handle h, double min_rate, double max_rate, double out[5], int64_t *count) {
    if (!h || !out) return CMI_E_INVALID;
    double sums[5];
    if (int rc = cmi_eval_resident_sums(h, min_rate, max_rate, sums)) return rc;
    const double cnt = sums[4];
    const double mae = sums[0] / cnt;
    out[0] = mae;
    out[1] = std::sqrt(sums[1] / cnt);
    out[2] = mae / (max_rate - min_rate);
    out[3] = sums[2] / cnt;
    out[4] = std::sqrt(sums[3] / cnt);
    if (count) *count = (int64_t)cnt;
    return CMI_OK;
}

extern "C" int cmi_level_schedule(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items,
                                  int order, int32_t *perm, int64_t *level_off, int64_t level_cap,
                                  int64_t *n_levels) {
    if (n < 0 || (n > 0 && (!u || !j)) || n_users <= 0 || n_items <= 0 || !n_levels || order < 0 || order > 3)
        return CMI_E_INVALID;
    for (int64_t t = 0; t < n; ++t)
        if (u[t] < 0 || u[t] >= n_users || j[t] < 0 || j[t] >= n_items) return CMI_E_INVALID;
    LevelSchedule sch;
    if (!build_level_schedule(n, u, j, n_users, n_items, order, sch)) return CMI_E_UNSUPPORTED;
    *n_levels = sch.n_levels();
    if (level_off) {
        if (level_cap < sch.n_levels() + 1) return CMI_E_INVALID;
        for (size_t i = 0; i < sch.level_off.size(); ++i) level_off[i] = sch.level_off[i];
    }
    if (perm)
        for (int64_t s = 0; s < n; ++s) perm[s] = sch.perm[(size_t)s];
    return CMI_OK;
}

extern "C" int cmi_chain_schedule(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int hub,
                                  int max_chain, int32_t *perm, int32_t *unit_off, int64_t unit_cap, int64_t *level_off,
                                  int64_t level_cap, int64_t *n_units, int64_t *n_levels, int *hub_used) {
    if (n < 0 || (n > 0 && (!u || !j)) || n_users <= 0 || n_items <= 0 || !n_units || !n_levels || max_chain < 1) return CMI_E_INVALID;
    for (int64_t t = 0; t < n; ++t)
        if (u[t] < 0 || u[t] >= n_users || j[t] < 0 || j[t] >= n_items) return CMI_E_INVALID;
    ChainSchedule cs;
    if (!build_chain_schedule(n, u, j, n_users, n_items, hub, max_chain, cs)) return CMI_E_UNSUPPORTED;
    *n_units = cs.n_units();
    *n_levels = cs.n_levels();
    if (hub_used) *hub_used = cs.hub_is_item;
    if (!perm) return CMI_OK;
    if (!unit_off || !level_off || unit_cap < cs.n_units() + 1 || level_cap < cs.n_levels() + 1) return CMI_E_INVALID;
    std::copy(cs.perm.begin(), cs.perm.end(), perm);
    std::copy(cs.unit_off.begin(), cs.unit_off.end(), unit_off);
    std::copy(cs.level_off.begin(), cs.level_off.end(), level_off);
    return CMI_OK;
}

// the same schedule built on the device (what cmi_set_ratings uses for large sets): tests compare the two element for element
extern "C" int cmi_chain_schedule_device(int device, int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int hub,
                                         int max_chain, int32_t *perm, int32_t *unit_off, int64_t unit_cap, int64_t *level_off,
                                         int64_t level_cap, int64_t *n_units, int64_t *n_levels, int *hub_used) {
    if (n < 0 || (n > 0 && (!u || !j)) || n_users <= 0 || n_items <= 0 || !n_units || !n_levels || max_chain < 1) return CMI_E_INVALID;
    if (device < 0 || device >= cmi_device_count()) return CMI_E_NO_DEVICE;
    for (int64_t t = 0; t < n; ++t)
        if (u[t] < 0 || u[t] >= n_users || j[t] < 0 || j[t] >= n_items) return CMI_E_INVALID;
    ChainSchedule cs;
    hipStream_t st = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return CMI_E_HIP;
    const bool ok = build_chain_schedule_device(device, st, n, u, j, n_users, n_items, hub, max_chain, cs);
    (void)hipStreamDestroy(st);
    if (!ok) return CMI_E_UNSUPPORTED;
    *n_units = cs.n_units();
    *n_levels = cs.n_levels();
    if (hub_used) *hub_used = cs.hub_is_item;
    if (!perm) return CMI_OK;
    if (!unit_off || !level_off || unit_cap < cs.n_units() + 1 || level_cap < cs.n_levels() + 1) return CMI_E_INVALID;
    std::copy(cs.perm.begin(), cs.perm.end(), perm);
    std::copy(cs.unit_off.begin(), cs.unit_off.end(), unit_off);
    std::copy(cs.level_off.begin(), cs.level_off.end(), level_off);
    return CMI_OK;
}

extern "C" int cmi_owner_schedule(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items, int hub, int n_owners,
                                  int depth, int32_t *perm, int64_t *own_off, uint32_t *want, uint32_t *flags, int *hub_used) {
    if (n < 0 || (n > 0 && (!u || !j)) || n_users <= 0 || n_items <= 0 || n_owners < 1 || depth < 1 || !perm || !own_off || !want || !flags)
        return CMI_E_INVALID;
    for (int64_t t = 0; t < n; ++t)
        if (u[t] < 0 || u[t] >= n_users || j[t] < 0 || j[t] >= n_items) return CMI_E_INVALID;
    OwnerSchedule os;
    if (!build_owner_schedule(n, u, j, n_users, n_items, hub, n_owners, depth, os)) return CMI_E_UNSUPPORTED;
    if (hub_used) *hub_used = os.hub_is_item;
    std::copy(os.perm.begin(), os.perm.end(), perm);
    std::copy(os.own_off.begin(), os.own_off.end(), own_off);
    std::copy(os.want.begin(), os.want.end(), want);
    std::copy(os.flags.begin(), os.flags.end(), flags);
    return CMI_OK;
}

// host-only views of the two schedule post-passes (tests): narrow runs of a level schedule, conflict-free CRS blocks
extern "C" int cmi_narrow_runs(int64_t n_levels, const int64_t *level_off, int64_t max_tuples, int64_t min_levels,
                               int32_t *run_len, int64_t *n_launches) {
    if (n_levels < 0 || !level_off || !n_launches || (n_levels > 0 && !run_len)) return CMI_E_INVALID;
    std::vector<int64_t> off(level_off, level_off + n_levels + 1);
    std::vector<int32_t> rl;
    *n_launches = build_narrow_runs(off, max_tuples, min_levels, rl);
    for (int64_t l = 0; l < n_levels; ++l) run_len[l] = rl[(size_t)l];
    return CMI_OK;
}

extern "C" int cmi_conflict_free_blocks(int64_t n, const int32_t *u, const int32_t *j, int32_t n_users, int32_t n_items,
                                        int32_t max_block, int32_t *off, int64_t off_cap, int64_t *n_blocks) {
    if (n < 0 || (n > 0 && (!u || !j)) || n_users <= 0 || n_items <= 0 || max_block <= 0 || !n_blocks || n >= ((int64_t)1 << 31))
        return CMI_E_INVALID;
    for (int64_t t = 0; t < n; ++t)
        if (u[t] < 0 || u[t] >= n_users || j[t] < 0 || j[t] >= n_items) return CMI_E_INVALID;
    std::vector<int32_t> o;
    build_conflict_free_blocks(n, u, j, n_users, n_items, max_block, o);
    *n_blocks = (int64_t)o.size() - 1;
    if (!off) return CMI_OK;
    if (off_cap < (int64_t)o.size()) return CMI_E_INVALID;
    std::copy(o.begin(), o.end(), off);
    return CMI_OK;
}

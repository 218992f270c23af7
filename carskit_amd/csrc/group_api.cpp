// group_api.cpp -- cmi_group_*: ONE handle that trains one recommender over several GPUs of a node from a single host process
// (the Java / C++ hosts have one process; carskit_amd/dist.py is the one-process-per-GPU form of the same algorithm).
//
// The reference path is a single sequential loop (e.g. CAMF_CI.java:79-123) and shards nowhere; its only parallelism is one
// thread per fold (CARSKit.java:395-412).  The algorithm shards BY USER: P[u], userBias[u], ucBias[u,:] are touched
// by exactly one shard; the item-side containers (Q, itemBias, icBias) are replicated and merged once per epoch:
//
//     every shard:  local order-exact epoch over its users' ratings             (cmi_train_epoch_async, concurrently)
//     exchange:     bucket_s = item_side_s - snapshot   (cmi_exchange_pack)
//                   reduce-scatter + all-gather of the buckets (RCCL over xGMI: ncclReduceScatter / ncclAllGather, one grouped
//                   call over all communicators of this process, on the instances' own streams)
//                   item_side = snapshot + bucket / W   (cmi_exchange_apply: the MEAN of the shards' moves)
//     loss:         all-reduce (sum) of the fp64 epoch losses -> what isConverged()/updateLRate() steer by
//
// Shards that share a device (a group of N on ONE GPU: the form the tests run on a single-GPU box) or CMI_GROUP_NO_RCCL=1 use
// the in-process exchange instead: the buckets are summed in shard order on shard 0's stream and copied back (peer copies) --
// the same arithmetic, no communicator.
#include "../../include/carskit_mi355x.h"
#include "env_knobs.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "cmi_instance.hpp"

struct cmi_group {
    int model = 0, k = 0, n_users = 0, n_items = 0, n_conds = 0;
    unsigned flags = 0;
    bool f64 = false;
    std::vector<int> dev;            // device of every shard
    std::vector<cmi_handle> inst;    // created by cmi_group_set_ratings (their user counts depend on the cut)
    std::vector<int32_t> cut;        // W + 1 user-id boundaries
    std::vector<int64_t> shard_n;    // tuples per shard
    std::vector<void *> bucket;
    std::vector<hipEvent_t> ev;      // per shard; ev[W] = shard 0's "sum ready"
    std::vector<hipEvent_t> tx;      // timing: tx[2s] / tx[2s+1] on shard s's stream before its pack / after its apply
    bool timed = false;
    int64_t x_count = 0;
    bool rccl = false;
    std::string path_note;           // which exchange runs and why (cmi_group_exchange_path): RCCL verified by the pre-flight, or the fallback's reason
    std::vector<ncclComm_t> comm;
    void *d_stage = nullptr;         // in-process exchange: staging buffer on shard 0's device
    double lr_scale = 1.0;           // local learning rate = lrate x lr_scale (cmi_group_set_lr_scale)
    double hp[5] = {0, 0, 0, 0, 0};  // regU regI regB regC globalMean
    bool have_hp = false;
    std::string err;
    float last_ms = 0.f;
};

static thread_local std::string g_group_create_err;

#define GRP_FAIL(g, code, ...)                    \
    do {                                          \
        char buf_[512];                           \
        snprintf(buf_, sizeof buf_, __VA_ARGS__); \
        (g)->err = buf_;                          \
        return (code);                            \
    } while (0)
#define GRP_HIP(g, expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) GRP_FAIL(g, CMI_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define GRP_NCCL(g, expr)                                                                                 \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess) GRP_FAIL(g, CMI_E_HIP, "%s failed: %s", #expr, ncclGetErrorString(r_));   \
    } while (0)
// a member call failed: carry its message
#define GRP_MEMBER(g, i, expr)                                                                                   \
    do {                                                                                                         \
        int rc_ = (expr);                                                                                        \
        if (rc_ != CMI_OK) GRP_FAIL(g, rc_, "shard %d (device %d): %s", (int)(i), (g)->dev[(size_t)(i)], cmi_last_error((g)->inst[(size_t)(i)])); \
    } while (0)

// THE exchange, one implementation for both hosts: the single-process group (cmi_group_*: ncclCommInitAll, one
// communicator per shard, grouped calls) and the one-process-per-GPU job (cmi_comm_*: ncclCommInitRank, carskit_amd/dist.py and
// bench.py --gpus N) issue exactly these three collectives on the instance's stream:
//   phase 0  reduce-scatter (sum) of the bucket of item-side moves: rank r receives slice r      |  xGMI is point to point: each rank's
//   phase 1  all-gather of the summed slices, in place                                           |  slice goes straight to its owner and
//   phase 2  all-reduce (sum) of the fp64 epoch loss                                             |  back, 2 x S/W per link pair, where a
//                                                                                                   ring all-reduce is bound by one link
// (CMI_DIST_ALLREDUCE=1: phase 0 = one all-reduce, phase 1 nothing -- the A/B of SURVEY 8e.)  In place: rank r's slice of its own
// bucket is the reduce-scatter output and the all-gather input.
static ncclResult_t exchange_collective(int phase, ncclComm_t comm, void *bucket, int64_t count, int world, int rank, bool f64, double *dloss,
                                        hipStream_t st) {
    static const bool allreduce = getenv("CMI_DIST_ALLREDUCE") != nullptr;
    const size_t es = f64 ? 8 : 4, chunk = (size_t)(count / world);
    const ncclDataType_t dt = f64 ? ncclDouble : ncclFloat;
    char *mine = (char *)bucket + (size_t)rank * chunk * es;
    switch (phase) {
    case 0: return allreduce ? ncclAllReduce(bucket, bucket, (size_t)count, dt, ncclSum, comm, st) : ncclReduceScatter(bucket, mine, chunk, dt, ncclSum, comm, st);
    case 1: return allreduce ? ncclSuccess : ncclAllGather(mine, bucket, chunk, dt, comm, st);
    default: return ncclAllReduce(dloss, dloss, 1, ncclDouble, ncclSum, comm, st);
    }
}

static bool user_side(int which) { return which == CMI_STATE_P || which == CMI_STATE_USER_BIAS || which == CMI_STATE_UC_BIAS; }

template <typename T>
__global__ void group_add_kernel(T *__restrict__ acc, const T *__restrict__ x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += x[i];
}
static hipError_t group_add(void *acc, const void *x, int64_t n, bool f64, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (f64) hipLaunchKernelGGL(group_add_kernel<double>, dim3(blocks), dim3(256), 0, s, (double *)acc, (const double *)x, n);
    else hipLaunchKernelGGL(group_add_kernel<float>, dim3(blocks), dim3(256), 0, s, (float *)acc, (const float *)x, n);
    return hipGetLastError();
}

static int group_preflight(cmi_group *g);

extern "C" const char *cmi_group_exchange_path(cmi_group_handle g) { return g ? g->path_note.c_str() : ""; }

extern "C" const char *cmi_group_last_error(cmi_group_handle g) { return g ? g->err.c_str() : g_group_create_err.c_str(); }

static void group_free_instances(cmi_group *g) {
    for (ncclComm_t c : g->comm)
        if (c) ncclCommDestroy(c);
    g->comm.clear();
    if (g->d_stage) {
        hipSetDevice(g->dev[0]);
        hipFree(g->d_stage);
        g->d_stage = nullptr;
    }
    for (size_t i = 0; i < g->tx.size(); ++i)
        if (g->tx[i]) {
            hipSetDevice(g->dev[i / 2]);
            hipEventDestroy(g->tx[i]);
        }
    g->tx.clear();
    g->timed = false;
    for (size_t i = 0; i < g->ev.size(); ++i)
        if (g->ev[i]) {
            hipSetDevice(g->dev[i + 1 == g->ev.size() ? 0 : i]);
            hipEventDestroy(g->ev[i]);
        }
    g->ev.clear();
    for (cmi_handle h : g->inst)
        if (h) cmi_destroy(h);
    g->inst.clear();
    g->bucket.clear();
    g->cut.clear();
    g->shard_n.clear();
}

extern "C" int cmi_group_destroy(cmi_group_handle g) {
    if (!g) return CMI_OK;
    group_free_instances(g);
    delete g;
    return CMI_OK;
}

extern "C" int cmi_group_create(int model, int k, int n_users, int n_items, int n_conds, int n_shards, const int *devices, unsigned flags,
                                cmi_group_handle *out) {
    if (out) *out = nullptr;
    if (!out || n_shards < 1 || n_shards > 64 || k <= 0 || n_users < n_shards || n_items <= 0 || n_conds < 0) {
        g_group_create_err = "cmi_group_create: invalid argument (1..64 shards, at least one user per shard)";
        return CMI_E_INVALID;
    }
    if (model == CMI_MODEL_CAMF_C || (model >= CMI_MODEL_SVDPP && model <= CMI_MODEL_CAMF_MCS)) {
        if (n_shards > 1) {
            g_group_create_err = "cmi_group_create: CAMF_C, SVD++ and CAMF_ICS/LCS/MCS are single serial chains (every tuple updates "
                                 "parameters every other tuple reads) and are not sharded; use one shard";
            return CMI_E_UNSUPPORTED;
        }
    }
    const int ndev = cmi_device_count();
    if (ndev <= 0) {
        g_group_create_err = "cmi_group_create: no HIP device visible (libcarskit_mi355x has no CPU fallback)";
        return CMI_E_NO_DEVICE;
    }
    cmi_group *g = new cmi_group();
    g->model = model, g->k = k, g->n_users = n_users, g->n_items = n_items, g->n_conds = n_conds, g->flags = flags;
    g->f64 = flags & CMI_FLAG_STATE_F64;
    for (int s = 0; s < n_shards; ++s) {
        const int d = devices ? devices[s] : s % ndev; // default: round robin over the visible devices
        if (d < 0 || d >= ndev) {
            g_group_create_err = "cmi_group_create: device index out of range";
            delete g;
            return CMI_E_INVALID;
        }
        g->dev.push_back(d);
    }
    *out = g;
    return CMI_OK;
}

extern "C" int cmi_group_size(cmi_group_handle g) { return g ? (int)g->dev.size() : 0; }

extern "C" int cmi_group_set_hparams(cmi_group_handle g, double regU, double regI, double regB, double regC, double global_mean) {
    if (!g) return CMI_E_INVALID;
    g->hp[0] = regU, g->hp[1] = regI, g->hp[2] = regB, g->hp[3] = regC, g->hp[4] = global_mean;
    g->have_hp = true;
    for (size_t i = 0; i < g->inst.size(); ++i) GRP_MEMBER(g, i, cmi_set_hparams(g->inst[i], regU, regI, regB, regC, global_mean));
    return CMI_OK;
}

// Contiguous user-id ranges cut so that every shard holds about n / W tuples (the rule of carskit_amd/dist.py shard_by_user:
// boundary r = first user at which the running tuple count reaches n * r / W, at least one user per shard).
static void cut_users(int64_t n, const int32_t *u, int32_t n_users, int W, std::vector<int32_t> &cut) {
    std::vector<int64_t> cum((size_t)n_users + 1, 0);
    for (int64_t t = 0; t < n; ++t) cum[(size_t)u[t] + 1]++;
    for (int32_t x = 0; x < n_users; ++x) cum[(size_t)x + 1] += cum[(size_t)x];
    cut.assign((size_t)W + 1, 0);
    for (int r = 1; r < W; ++r) {
        const double target = (double)n * r / W;
        // first index c with cum[c] >= target   (numpy.searchsorted(cum, target, side="left"))
        int32_t c = (int32_t)(std::lower_bound(cum.begin(), cum.end(), target, [](int64_t a, double b) { return (double)a < b; }) - cum.begin());
        c = std::max(c, cut[(size_t)r - 1] + 1);
        c = std::min(c, n_users - (W - r));
        cut[(size_t)r] = c;
    }
    cut[(size_t)W] = n_users;
}

static int shard_of(const std::vector<int32_t> &cut, int32_t user) {
    return (int)(std::upper_bound(cut.begin() + 1, cut.end(), user) - cut.begin() - 1);
}

extern "C" int cmi_group_set_ratings(cmi_group_handle g, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, const double *r,
                                     int32_t n_ctx, const int32_t *ctx_ptr, const int32_t *ctx_conds) {
    if (!g) return CMI_E_INVALID;
    if (n < 0 || (n > 0 && (!u || !j || !r))) GRP_FAIL(g, CMI_E_INVALID, "group_set_ratings: null tuple arrays");
    const int W = (int)g->dev.size();
    for (int64_t t = 0; t < n; ++t)
        if (u[t] < 0 || u[t] >= g->n_users) GRP_FAIL(g, CMI_E_INVALID, "group_set_ratings: user id %d out of range at tuple %lld", u[t], (long long)t);
    group_free_instances(g);
    cut_users(n, u, g->n_users, W, g->cut);
    // tuples of every shard, CRS order kept, user ids re-based to the shard
    std::vector<std::vector<int32_t>> su((size_t)W), sj((size_t)W), sc((size_t)W);
    std::vector<std::vector<double>> sr((size_t)W);
    g->shard_n.assign((size_t)W, 0);
    for (int64_t t = 0; t < n; ++t) g->shard_n[(size_t)shard_of(g->cut, u[t])]++;
    for (int s = 0; s < W; ++s) {
        su[(size_t)s].reserve((size_t)g->shard_n[(size_t)s]);
        sj[(size_t)s].reserve((size_t)g->shard_n[(size_t)s]);
        if (ctx) sc[(size_t)s].reserve((size_t)g->shard_n[(size_t)s]);
        sr[(size_t)s].reserve((size_t)g->shard_n[(size_t)s]);
    }
    for (int64_t t = 0; t < n; ++t) {
        const int s = shard_of(g->cut, u[t]);
        su[(size_t)s].push_back(u[t] - g->cut[(size_t)s]);
        sj[(size_t)s].push_back(j[t]);
        if (ctx) sc[(size_t)s].push_back(ctx[t]);
        sr[(size_t)s].push_back(r[t]);
    }
    g->inst.assign((size_t)W, nullptr);
    for (int s = 0; s < W; ++s) {
        const int rc = cmi_create(g->model, g->k, g->cut[(size_t)s + 1] - g->cut[(size_t)s], g->n_items, g->n_conds, g->dev[(size_t)s], g->flags,
                                  &g->inst[(size_t)s]);
        if (rc != CMI_OK) GRP_FAIL(g, rc, "shard %d (device %d): %s", s, g->dev[(size_t)s], cmi_last_error(nullptr));
        if (g->have_hp) GRP_MEMBER(g, s, cmi_set_hparams(g->inst[(size_t)s], g->hp[0], g->hp[1], g->hp[2], g->hp[3], g->hp[4]));
    }
    // schedule construction is host integer work (seconds for tens of millions of tuples): one thread per shard
    std::vector<int> rcs((size_t)W, CMI_OK);
    {
        std::vector<std::thread> th;
        for (int s = 0; s < W; ++s)
            th.emplace_back([&, s] {
                rcs[(size_t)s] = cmi_set_ratings(g->inst[(size_t)s], g->shard_n[(size_t)s], su[(size_t)s].data(), sj[(size_t)s].data(),
                                                 ctx ? sc[(size_t)s].data() : nullptr, sr[(size_t)s].data(), n_ctx, ctx_ptr, ctx_conds);
            });
        for (std::thread &t : th) t.join();
    }
    for (int s = 0; s < W; ++s)
        if (rcs[(size_t)s] != CMI_OK) GRP_FAIL(g, rcs[(size_t)s], "shard %d (device %d): %s", s, g->dev[(size_t)s], cmi_last_error(g->inst[(size_t)s]));
    if (W == 1) return CMI_OK;

    // exchange plumbing
    g->bucket.assign((size_t)W, nullptr);
    for (int s = 0; s < W; ++s) {
        int64_t cnt = 0;
        GRP_MEMBER(g, s, cmi_exchange_setup(g->inst[(size_t)s], W, &g->bucket[(size_t)s], &cnt));
        if (s == 0) g->x_count = cnt;
        else if (cnt != g->x_count) GRP_FAIL(g, CMI_E_INVALID, "group_set_ratings: bucket sizes differ between shards");
    }
    bool distinct = true;
    for (int a = 0; a < W; ++a)
        for (int b = a + 1; b < W; ++b) distinct = distinct && g->dev[(size_t)a] != g->dev[(size_t)b];
    g->ev.assign((size_t)W + 1, nullptr);
    for (int s = 0; s <= W; ++s) { // ev[W] belongs to shard 0's device
        GRP_HIP(g, hipSetDevice(g->dev[s == W ? 0 : (size_t)s]));
        GRP_HIP(g, hipEventCreateWithFlags(&g->ev[(size_t)s], hipEventDisableTiming));
    }
    g->tx.assign((size_t)W * 2, nullptr);
    for (int s = 0; s < 2 * W; ++s) {
        GRP_HIP(g, hipSetDevice(g->dev[(size_t)s / 2]));
        GRP_HIP(g, hipEventCreate(&g->tx[(size_t)s]));
    }
    // The in-process exchange (peer copies) is ALWAYS set up: it is what shards that share a device use, and what a group falls back
    // to when RCCL cannot be initialised or does not pass the pre-flight on this node (the first N > 1 RCCL call
    // ever executed must not be the timed one, and a failure must still leave a working trainer).
    GRP_HIP(g, hipSetDevice(g->dev[0]));
    GRP_HIP(g, hipMalloc(&g->d_stage, (size_t)g->x_count * (g->f64 ? 8 : 4) + 8));
    for (int s = 1; s < W; ++s)
        if (g->dev[(size_t)s] != g->dev[0]) {
            int can = 0;
            hipDeviceCanAccessPeer(&can, g->dev[0], g->dev[(size_t)s]);
            if (can) {
                hipSetDevice(g->dev[0]);
                hipDeviceEnablePeerAccess(g->dev[(size_t)s], 0); // already enabled is fine
                hipSetDevice(g->dev[(size_t)s]);
                hipDeviceEnablePeerAccess(g->dev[0], 0);
                (void)hipGetLastError();
            }
        }
    g->rccl = false;
    g->path_note = "in-process exchange (sums on shard 0's stream, peer copies)";
    // CMI_GROUP_TRY_RCCL=1 (test hook): attempt RCCL although shards share a device -- ncclCommInitAll refuses duplicate devices, which
    // drives the fallback below on a single-GPU box (tests/test_gpu_bench_group.py)
    const bool try_rccl = (distinct || getenv("CMI_GROUP_TRY_RCCL")) && !cmi_exp_env("CMI_GROUP_NO_RCCL");
    if (!distinct && !try_rccl) g->path_note += ": shards share a device";
    if (try_rccl) {
        g->comm.assign((size_t)W, nullptr);
        const ncclResult_t r = ncclCommInitAll(g->comm.data(), W, g->dev.data());
        if (r != ncclSuccess) {
            g->comm.clear();
            (void)hipGetLastError();
            g->path_note = std::string("in-process exchange (peer copies) -- FALLBACK: ncclCommInitAll failed: ") + ncclGetErrorString(r);
            fprintf(stderr, "[cmi] group: %s\n", g->path_note.c_str());
        } else {
            g->rccl = true;
            if (int rc = group_preflight(g)) return rc; // (may switch to the in-process exchange; only a HIP failure is an error)
        }
    }
    return CMI_OK;
}

static int need_ratings(cmi_group *g, const char *what) {
    if (g->inst.empty()) GRP_FAIL(g, CMI_E_INVALID, "%s: call cmi_group_set_ratings first (the user cut depends on the ratings)", what);
    return CMI_OK;
}

extern "C" int cmi_group_set_state(cmi_group_handle g, int which, const void *src, int64_t count, int dtype) {
    if (!g || !src) return CMI_E_INVALID;
    if (int rc = need_ratings(g, "group_set_state")) return rc;
    const int W = (int)g->inst.size();
    const size_t es = dtype == CMI_DTYPE_F64 ? 8 : 4;
    if (!user_side(which)) {
        for (int s = 0; s < W; ++s) GRP_MEMBER(g, s, cmi_set_state(g->inst[(size_t)s], which, src, count, dtype));
        return CMI_OK;
    }
    if (count % g->n_users != 0) GRP_FAIL(g, CMI_E_INVALID, "group_set_state: container %d: %lld elements is not a multiple of %d users", which, (long long)count, g->n_users);
    const int64_t cols = count / g->n_users;
    for (int s = 0; s < W; ++s) {
        const int64_t lo = g->cut[(size_t)s], hi = g->cut[(size_t)s + 1];
        GRP_MEMBER(g, s, cmi_set_state(g->inst[(size_t)s], which, (const char *)src + (size_t)(lo * cols) * es, (hi - lo) * cols, dtype));
    }
    return CMI_OK;
}

extern "C" int cmi_group_get_state(cmi_group_handle g, int which, void *dst, int64_t count, int dtype) {
    if (!g || !dst) return CMI_E_INVALID;
    if (int rc = need_ratings(g, "group_get_state")) return rc;
    const int W = (int)g->inst.size();
    const size_t es = dtype == CMI_DTYPE_F64 ? 8 : 4;
    if (!user_side(which)) { // replicated: every shard holds the merged value
        GRP_MEMBER(g, 0, cmi_get_state(g->inst[0], which, dst, count, dtype));
        return CMI_OK;
    }
    if (count % g->n_users != 0) GRP_FAIL(g, CMI_E_INVALID, "group_get_state: container %d: %lld elements is not a multiple of %d users", which, (long long)count, g->n_users);
    const int64_t cols = count / g->n_users;
    for (int s = 0; s < W; ++s) {
        const int64_t lo = g->cut[(size_t)s], hi = g->cut[(size_t)s + 1];
        GRP_MEMBER(g, s, cmi_get_state(g->inst[(size_t)s], which, (char *)dst + (size_t)(lo * cols) * es, (hi - lo) * cols, dtype));
    }
    return CMI_OK;
}

// The collective part of the exchange over the first `count` elements of every shard's bucket (count: a multiple of W) and the fp64
// loss words dl[s], enqueued on the shards' streams st[s]: through RCCL (grouped calls over the process's communicators) or in process.
static int exchange_buckets(cmi_group *g, bool rccl, int64_t count, const std::vector<hipStream_t> &st, const std::vector<double *> &dl) {
    const int W = (int)g->inst.size();
    const size_t es = g->f64 ? 8 : 4;
    if (rccl) {
        // every collective of the exchange is one grouped call over all communicators of this process
        for (int phase = 0; phase < 3; ++phase) {
            GRP_NCCL(g, ncclGroupStart());
            for (int s = 0; s < W; ++s) {
                const ncclResult_t r = exchange_collective(phase, g->comm[(size_t)s], g->bucket[(size_t)s], count, W, s, g->f64, dl[(size_t)s], st[(size_t)s]);
                if (r != ncclSuccess) {
                    (void)ncclGroupEnd();
                    GRP_FAIL(g, CMI_E_HIP, "exchange (phase %d, shard %d) failed: %s", phase, s, ncclGetErrorString(r));
                }
            }
            GRP_NCCL(g, ncclGroupEnd());
        }
    } else {
        // in-process: shard 0's stream waits for every shard's pack, sums the buckets (and the losses) in shard order, the others copy
        // the sums back on their own streams
        for (int s = 1; s < W; ++s) {
            GRP_HIP(g, hipSetDevice(g->dev[(size_t)s]));
            GRP_HIP(g, hipEventRecord(g->ev[(size_t)s], st[(size_t)s]));
        }
        GRP_HIP(g, hipSetDevice(g->dev[0]));
        double *stage_loss = (double *)((char *)g->d_stage + (size_t)count * es);
        for (int s = 1; s < W; ++s) {
            GRP_HIP(g, hipStreamWaitEvent(st[0], g->ev[(size_t)s], 0));
            const void *src = g->bucket[(size_t)s];
            if (g->dev[(size_t)s] != g->dev[0]) {
                GRP_HIP(g, hipMemcpyPeerAsync(g->d_stage, g->dev[0], g->bucket[(size_t)s], g->dev[(size_t)s], (size_t)count * es, st[0]));
                GRP_HIP(g, hipMemcpyPeerAsync(stage_loss, g->dev[0], dl[(size_t)s], g->dev[(size_t)s], 8, st[0]));
                src = g->d_stage;
                GRP_HIP(g, group_add(dl[0], stage_loss, 1, true, st[0]));
            } else {
                GRP_HIP(g, group_add(dl[0], dl[(size_t)s], 1, true, st[0]));
            }
            GRP_HIP(g, group_add(g->bucket[0], src, count, g->f64, st[0]));
        }
        GRP_HIP(g, hipEventRecord(g->ev[(size_t)W], st[0]));
        for (int s = 1; s < W; ++s) {
            GRP_HIP(g, hipSetDevice(g->dev[(size_t)s]));
            GRP_HIP(g, hipStreamWaitEvent(st[(size_t)s], g->ev[(size_t)W], 0));
            GRP_HIP(g, hipMemcpyPeerAsync(g->bucket[(size_t)s], g->dev[(size_t)s], g->bucket[0], g->dev[0], (size_t)count * es, st[(size_t)s]));
            GRP_HIP(g, hipMemcpyPeerAsync(dl[(size_t)s], g->dev[(size_t)s], dl[0], g->dev[0], 8, st[(size_t)s]));
            GRP_HIP(g, hipEventRecord(g->ev[(size_t)s], st[(size_t)s])); // shard 0 must not start its next pack before the copies have read bucket 0
        }
        GRP_HIP(g, hipSetDevice(g->dev[0]));
        for (int s = 1; s < W; ++s) GRP_HIP(g, hipStreamWaitEvent(st[0], g->ev[(size_t)s], 0));
    }
    return CMI_OK;
}

// Pre-flight of a group that is about to use RCCL: one small exchange of exactly representable values (integers: any order of the
// additions gives the same bits) through the RCCL path AND through the in-process path, each compared with the sums computed on the
// host and so with each other, bit for bit.  It also takes RCCL's first-call cost (channel setup) out of the first timed epoch.  RCCL
// returning an error or a wrong sum switches the group to the in-process exchange, says so on stderr and in cmi_group_exchange_path.
static int group_preflight(cmi_group *g) {
    const int W = (int)g->inst.size();
    const size_t es = g->f64 ? 8 : 4;
    const int64_t count = (int64_t)W * std::min<int64_t>(g->x_count / W, 16384);
    if (count <= 0) return CMI_OK;
    std::vector<hipStream_t> st((size_t)W);
    std::vector<double *> dl((size_t)W);
    for (int s = 0; s < W; ++s) {
        GRP_MEMBER(g, s, cmi_stream(g->inst[(size_t)s], (void **)&st[(size_t)s]));
        GRP_MEMBER(g, s, cmi_loss_device_ptr(g->inst[(size_t)s], (void **)&dl[(size_t)s]));
    }
    auto value = [](int64_t i, int s) { return (double)((i * 7 + (int64_t)s * 13) % 97 - 48); };
    std::vector<double> want((size_t)count, 0.0);
    for (int s = 0; s < W; ++s)
        for (int64_t i = 0; i < count; ++i) want[(size_t)i] += value(i, s);
    const double want_loss = 0.5 * W * (W + 1);
    auto run = [&](bool rccl, std::string &why) -> int { // CMI_OK + why.empty(): verified; CMI_OK + why: this path is unusable
        std::vector<char> host((size_t)count * es);
        for (int s = 0; s < W; ++s) {
            for (int64_t i = 0; i < count; ++i) {
                if (g->f64) ((double *)host.data())[i] = value(i, s);
                else ((float *)host.data())[i] = (float)value(i, s);
            }
            const double l = s + 1.0;
            GRP_HIP(g, hipSetDevice(g->dev[(size_t)s]));
            GRP_HIP(g, hipMemcpyAsync(g->bucket[(size_t)s], host.data(), (size_t)count * es, hipMemcpyHostToDevice, st[(size_t)s]));
            GRP_HIP(g, hipMemcpyAsync(dl[(size_t)s], &l, 8, hipMemcpyHostToDevice, st[(size_t)s]));
            GRP_HIP(g, hipStreamSynchronize(st[(size_t)s])); // (host and l are reused)
        }
        const std::string keep = g->err;
        if (exchange_buckets(g, rccl, count, st, dl) != CMI_OK) {
            why = g->err;
            g->err = keep;
            (void)hipGetLastError();
            return CMI_OK;
        }
        for (int s = 0; s < W; ++s) {
            double l = 0.0;
            GRP_HIP(g, hipSetDevice(g->dev[(size_t)s]));
            GRP_HIP(g, hipMemcpyAsync(host.data(), g->bucket[(size_t)s], (size_t)count * es, hipMemcpyDeviceToHost, st[(size_t)s]));
            GRP_HIP(g, hipMemcpyAsync(&l, dl[(size_t)s], 8, hipMemcpyDeviceToHost, st[(size_t)s]));
            GRP_HIP(g, hipStreamSynchronize(st[(size_t)s]));
            for (int64_t i = 0; i < count && why.empty(); ++i) {
                const double got = g->f64 ? ((double *)host.data())[i] : (double)((float *)host.data())[i];
                if (got != want[(size_t)i]) {
                    char buf[160];
                    snprintf(buf, sizeof buf, "wrong sum on shard %d, element %lld: %.17g instead of %.17g", s, (long long)i, got, want[(size_t)i]);
                    why = buf;
                }
            }
            if (why.empty() && l != want_loss) why = "wrong loss sum";
        }
        return CMI_OK;
    };
    std::string why_rccl, why_local;
    if (int rc = run(true, why_rccl)) return rc;
    if (getenv("CMI_GROUP_PREFLIGHT_FAIL")) why_rccl = "forced by CMI_GROUP_PREFLIGHT_FAIL (test hook)";
    if (int rc = run(false, why_local)) return rc;
    if (!why_local.empty()) GRP_FAIL(g, CMI_E_HIP, "group: the in-process exchange failed its pre-flight: %s", why_local.c_str());
    if (why_rccl.empty()) {
        g->path_note = "RCCL (grouped ncclReduceScatter + ncclAllGather + loss ncclAllReduce over the process's communicators); pre-flight: "
                       "bit-identical to the in-process exchange and to the host's sums";
    } else {
        for (ncclComm_t c : g->comm)
            if (c) ncclCommAbort(c);
        g->comm.clear();
        g->rccl = false;
        g->path_note = "in-process exchange (peer copies) -- FALLBACK: RCCL pre-flight: " + why_rccl;
        fprintf(stderr, "[cmi] group: %s\n", g->path_note.c_str());
    }
    return CMI_OK;
}

// the epoch-boundary merge of the item-side containers + the global loss; everything enqueued on the shards' streams
static int group_exchange(cmi_group *g) {
    const int W = (int)g->inst.size();
    std::vector<hipStream_t> st((size_t)W);
    std::vector<double *> dl((size_t)W);
    for (int s = 0; s < W; ++s) {
        GRP_MEMBER(g, s, cmi_stream(g->inst[(size_t)s], (void **)&st[(size_t)s]));
        GRP_HIP(g, hipSetDevice(g->dev[(size_t)s]));
        GRP_HIP(g, hipEventRecord(g->tx[(size_t)2 * s], st[(size_t)s]));
        GRP_MEMBER(g, s, cmi_exchange_pack(g->inst[(size_t)s]));
        GRP_MEMBER(g, s, cmi_loss_device_ptr(g->inst[(size_t)s], (void **)&dl[(size_t)s]));
    }
    if (int rc = exchange_buckets(g, g->rccl, g->x_count, st, dl)) return rc;
    for (int s = 0; s < W; ++s) {
        GRP_MEMBER(g, s, cmi_exchange_apply(g->inst[(size_t)s], 1.0 / W));
        GRP_HIP(g, hipSetDevice(g->dev[(size_t)s]));
        GRP_HIP(g, hipEventRecord(g->tx[(size_t)2 * s + 1], st[(size_t)s]));
    }
    g->timed = true;
    return CMI_OK;
}

// HIP-event times of the most recent epoch, per shard: the local epoch's launches (cmi_last_epoch_ms of the member) and the exchange
// behind it (pack .. apply on the shard's stream: collectives or in-process sums, INCLUDING the wait for the slowest shard)
extern "C" int cmi_group_last_times(cmi_group_handle g, float *compute_ms, float *exchange_ms) {
    if (!g || !compute_ms || !exchange_ms) return CMI_E_INVALID;
    if (int rc = need_ratings(g, "group_last_times")) return rc;
    const int W = (int)g->inst.size();
    for (int s = 0; s < W; ++s) {
        GRP_MEMBER(g, s, cmi_last_epoch_ms(g->inst[(size_t)s], &compute_ms[s]));
        exchange_ms[s] = 0.f;
        if (W > 1 && g->timed) {
            GRP_HIP(g, hipSetDevice(g->dev[(size_t)s]));
            GRP_HIP(g, hipEventSynchronize(g->tx[(size_t)2 * s + 1]));
            GRP_HIP(g, hipEventElapsedTime(&exchange_ms[s], g->tx[(size_t)2 * s], g->tx[(size_t)2 * s + 1]));
        }
    }
    return CMI_OK;
}

extern "C" int cmi_group_train_epoch(cmi_group_handle g, double lrate, double *loss_out) {
    if (!g) return CMI_E_INVALID;
    if (int rc = need_ratings(g, "group_train_epoch")) return rc;
    const int W = (int)g->inst.size();
    const double local_lr = W > 1 ? lrate * g->lr_scale : lrate;
    for (int s = 0; s < W; ++s) GRP_MEMBER(g, s, cmi_train_epoch_async(g->inst[(size_t)s], local_lr));
    if (W > 1)
        if (int rc = group_exchange(g)) return rc;
    // the one host synchronisation of the epoch: the (already global) loss of shard 0, then the other shards' streams
    double loss = 0.0, other = 0.0;
    GRP_MEMBER(g, 0, cmi_last_loss(g->inst[0], &loss));
    for (int s = 1; s < W; ++s) GRP_MEMBER(g, s, cmi_last_loss(g->inst[(size_t)s], &other)); // also surfaces a stalled owner epoch of that shard
    if (loss_out) *loss_out = loss;
    return CMI_OK;
}

// The mean merge divides every item row's move by W: at equal rate a W-shard run needs 1.2x / 1.4x / 1.6x the epochs of the sequential
// run for the same training RMSE (W = 2 / 4 / 8).  Scaling the LOCAL rate by sqrt(W) brings that to <= 1.2x, lr x W diverges at W = 8
// (tests/exp_merge_rule.py --time-to-rmse).  The host keeps steering the base rate (bold driver); default 1.
extern "C" int cmi_group_set_lr_scale(cmi_group_handle g, double scale) {
    if (!g) return CMI_E_INVALID;
    if (!(scale > 0.0)) GRP_FAIL(g, CMI_E_INVALID, "group_set_lr_scale: scale must be positive");
    g->lr_scale = scale;
    return CMI_OK;
}

extern "C" int cmi_group_train_from(cmi_group_handle g, int first_iter, double prev_loss, int num_iters, double init_lrate, double max_lrate,
                                    int bold_driver, double decay, int early_stop, double *losses, double *lrates, int *iters_run,
                                    double *final_lrate) {
    if (!g) return CMI_E_INVALID;
    std::string err;
    const int rc = cmi_train_loop([g](double lr, double *loss) { return cmi_group_train_epoch(g, lr, loss); }, err, first_iter, prev_loss, num_iters,
                                  init_lrate, max_lrate, bold_driver, decay, early_stop, losses, lrates, iters_run, final_lrate);
    if (rc != CMI_OK && !err.empty()) g->err = err;
    return rc;
}

extern "C" int cmi_group_train(cmi_group_handle g, int num_iters, double init_lrate, double max_lrate, int bold_driver, double decay,
                               int early_stop, double *losses, double *lrates, int *iters_run, double *final_lrate) {
    return cmi_group_train_from(g, 1, 0.0, num_iters, init_lrate, max_lrate, bold_driver, decay, early_stop, losses, lrates, iters_run, final_lrate);
}

// test tuples go to the shard that owns their user
struct Routed {
    std::vector<std::vector<int32_t>> u, j, c;
    std::vector<std::vector<double>> r;
    std::vector<std::vector<int64_t>> pos;
};
static int route(cmi_group *g, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, const double *r, Routed &o) {
    const size_t W = g->inst.size();
    o.u.assign(W, {}), o.j.assign(W, {}), o.c.assign(W, {}), o.r.assign(W, {}), o.pos.assign(W, {});
    for (int64_t t = 0; t < n; ++t) {
        if (u[t] < 0 || u[t] >= g->n_users) GRP_FAIL(g, CMI_E_INVALID, "eval: user id %d out of range at tuple %lld", u[t], (long long)t);
        const size_t s = (size_t)shard_of(g->cut, u[t]);
        o.u[s].push_back(u[t] - g->cut[s]);
        o.j[s].push_back(j[t]);
        if (ctx) o.c[s].push_back(ctx[t]);
        if (r) o.r[s].push_back(r[t]);
        o.pos[s].push_back(t);
    }
    return CMI_OK;
}

extern "C" int cmi_group_eval_ratings(cmi_group_handle g, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, const double *r,
                                      double min_rate, double max_rate, double *out, int64_t *count) {
    if (!g || !out) return CMI_E_INVALID;
    if (int rc = need_ratings(g, "group_eval_ratings")) return rc;
    if (n < 0 || (n > 0 && (!u || !j || !r))) GRP_FAIL(g, CMI_E_INVALID, "group_eval_ratings: null argument");
    Routed ro;
    if (int rc = route(g, n, u, j, ctx, r, ro)) return rc;
    double tot[5] = {0, 0, 0, 0, 0};
    for (size_t s = 0; s < g->inst.size(); ++s) {
        double sums[5];
        const int64_t m = (int64_t)ro.u[s].size();
        GRP_MEMBER(g, s, cmi_eval_sums(g->inst[s], m, ro.u[s].data(), ro.j[s].data(), ctx ? ro.c[s].data() : nullptr, ro.r[s].data(), min_rate, max_rate, sums));
        for (int c = 0; c < 5; ++c) tot[c] += sums[c];
    }
    const double cnt = tot[4], mae = tot[0] / cnt;
    out[0] = mae;
    out[1] = std::sqrt(tot[1] / cnt);
    out[2] = mae / (max_rate - min_rate);
    out[3] = tot[2] / cnt;
    out[4] = std::sqrt(tot[3] / cnt);
    if (count) *count = (int64_t)cnt;
    return CMI_OK;
}

extern "C" int cmi_group_predict_batch(cmi_group_handle g, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, int bound, double lo,
                                       double hi, double *out) {
    if (!g || !out) return CMI_E_INVALID;
    if (int rc = need_ratings(g, "group_predict_batch")) return rc;
    if (n < 0 || (n > 0 && (!u || !j))) GRP_FAIL(g, CMI_E_INVALID, "group_predict_batch: null argument");
    Routed ro;
    if (int rc = route(g, n, u, j, ctx, nullptr, ro)) return rc;
    std::vector<double> tmp;
    for (size_t s = 0; s < g->inst.size(); ++s) {
        const int64_t m = (int64_t)ro.u[s].size();
        if (m == 0) continue;
        tmp.resize((size_t)m);
        GRP_MEMBER(g, s, cmi_predict_batch(g->inst[s], m, ro.u[s].data(), ro.j[s].data(), ctx ? ro.c[s].data() : nullptr, bound, lo, hi, tmp.data()));
        for (int64_t q = 0; q < m; ++q) out[ro.pos[s][(size_t)q]] = tmp[(size_t)q];
    }
    return CMI_OK;
}

extern "C" int cmi_group_shard_info(cmi_group_handle g, int shard, int64_t info[6]) {
    if (!g || !info) return CMI_E_INVALID;
    if (int rc = need_ratings(g, "group_shard_info")) return rc;
    if (shard < 0 || shard >= (int)g->inst.size()) GRP_FAIL(g, CMI_E_INVALID, "group_shard_info: shard %d out of range", shard);
    info[0] = g->cut[(size_t)shard];
    info[1] = g->cut[(size_t)shard + 1];
    info[2] = g->shard_n[(size_t)shard];
    info[3] = g->dev[(size_t)shard];
    info[4] = g->inst.size() > 1 ? (g->rccl ? 1 : 2) : 0; // exchange: 0 none (one shard), 1 RCCL, 2 in-process
    info[5] = g->x_count;
    return CMI_OK;
}

extern "C" int cmi_group_member(cmi_group_handle g, int shard, cmi_handle *out) {
    if (!g || !out) return CMI_E_INVALID;
    if (int rc = need_ratings(g, "group_member")) return rc;
    if (shard < 0 || shard >= (int)g->inst.size()) GRP_FAIL(g, CMI_E_INVALID, "group_member: shard %d out of range", shard);
    *out = g->inst[(size_t)shard];
    return CMI_OK;
}

// ---- early stop on a measure for a sharded recommender (IterativeRecommender.java:149-161: isConverged() scores the test set after
// every epoch): the test tuples go to the shard that owns their user ONCE and stay on its device ----------------------------------------
extern "C" int cmi_group_set_eval_ratings(cmi_group_handle g, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, const double *r) {
    if (!g) return CMI_E_INVALID;
    if (int rc = need_ratings(g, "group_set_eval_ratings")) return rc;
    if (n < 0 || (n > 0 && (!u || !j || !r))) GRP_FAIL(g, CMI_E_INVALID, "group_set_eval_ratings: null argument");
    Routed ro;
    if (int rc = route(g, n, u, j, ctx, r, ro)) return rc;
    for (size_t s = 0; s < g->inst.size(); ++s)
        GRP_MEMBER(g, s, cmi_set_eval_ratings(g->inst[s], (int64_t)ro.u[s].size(), ro.u[s].data(), ro.j[s].data(), ctx ? ro.c[s].data() : nullptr,
                                              ro.r[s].data()));
    return CMI_OK;
}

extern "C" int cmi_group_eval_resident(cmi_group_handle g, double min_rate, double max_rate, double out[5], int64_t *count) {
    if (!g || !out) return CMI_E_INVALID;
    if (int rc = need_ratings(g, "group_eval_resident")) return rc;
    double tot[5] = {0, 0, 0, 0, 0};
    for (size_t s = 0; s < g->inst.size(); ++s) { // sums in shard order: deterministic
        if (g->inst[s]->n_eval <= 0) continue;    // a shard that owns none of the test users
        double sums[5];
        GRP_MEMBER(g, s, cmi_eval_resident_sums(g->inst[s], min_rate, max_rate, sums));
        for (int c = 0; c < 5; ++c) tot[c] += sums[c];
    }
    if (!(tot[4] > 0)) GRP_FAIL(g, CMI_E_INVALID, "group_eval_resident: call cmi_group_set_eval_ratings first");
    const double cnt = tot[4], mae = tot[0] / cnt;
    out[0] = mae;
    out[1] = std::sqrt(tot[1] / cnt);
    out[2] = mae / (max_rate - min_rate);
    out[3] = tot[2] / cnt;
    out[4] = std::sqrt(tot[3] / cnt);
    if (count) *count = (int64_t)cnt;
    return CMI_OK;
}

// ---- cmi_comm_*: the same exchange for the one-process-per-GPU form (carskit_amd/dist.py, bench.py --gpus N under
// torch.distributed.run).  The host ranks only have to share 128 bytes once (the RCCL unique id: rank 0 makes it, the host's own
// rendezvous -- torch.distributed, MPI, a file -- hands it to the others); from then on every epoch's exchange is issued by this
// library on the instance's stream, through exchange_collective above. -----------------------------------------------------------------
static_assert(sizeof(ncclUniqueId) == CMI_COMM_ID_BYTES, "CMI_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

extern "C" int cmi_comm_unique_id(void *id) {
    if (!id) return CMI_E_INVALID;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return CMI_E_HIP;
    memcpy(id, &u, sizeof u);
    return CMI_OK;
}

void cmi_comm_release(cmi_instance *h) {
    for (hipEvent_t *e : {&h->evx0, &h->evx1})
        if (*e) {
            (void)hipSetDevice(h->device);
            (void)hipEventDestroy(*e);
            *e = nullptr;
        }
    h->exchange_timed = false;
    if (h->comm) {
        (void)hipSetDevice(h->device);
        (void)ncclCommDestroy((ncclComm_t)h->comm);
        h->comm = nullptr;
    }
    h->comm_world = 0;
}

#define COMM_NCCL(h, expr)                                                                                \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess) CMI_FAIL(h, CMI_E_HIP, "%s failed: %s", #expr, ncclGetErrorString(r_));   \
    } while (0)

extern "C" int cmi_comm_init(cmi_handle h, const void *id, int rank, int world) {
    if (!h || !id || world < 1 || rank < 0 || rank >= world) return CMI_E_INVALID;
    cmi_comm_release(h);
    CMI_HIP(h, hipSetDevice(h->device));
    void *bucket = nullptr;
    int64_t cnt = 0;
    if (int rc = cmi_exchange_setup(h, world, &bucket, &cnt)) return rc; // also snapshots the item-side state
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t c = nullptr;
    COMM_NCCL(h, ncclCommInitRank(&c, world, u, rank));
    h->comm = c;
    h->comm_rank = rank;
    h->comm_world = world;
    return CMI_OK;
}

// pack -> reduce-scatter -> all-gather -> apply(scale) -> loss all-reduce, all on the instance's stream (no host synchronisation)
extern "C" int cmi_comm_exchange(cmi_handle h, double scale) {
    if (!h) return CMI_E_INVALID;
    if (!h->comm) CMI_FAIL(h, CMI_E_INVALID, "comm_exchange: call cmi_comm_init first");
    CMI_HIP(h, hipSetDevice(h->device));
    if (!h->evx0) {
        CMI_HIP(h, hipEventCreate(&h->evx0));
        CMI_HIP(h, hipEventCreate(&h->evx1));
    }
    CMI_HIP(h, hipEventRecord(h->evx0, h->stream));
    if (int rc = cmi_exchange_pack(h)) return rc;
    double *dl = nullptr;
    if (int rc = cmi_loss_device_ptr(h, (void **)&dl)) return rc;
    for (int phase = 0; phase < 2; ++phase)
        COMM_NCCL(h, exchange_collective(phase, (ncclComm_t)h->comm, h->d_xbucket, h->x_count, h->comm_world, h->comm_rank, h->f64, dl, h->stream));
    if (int rc = cmi_exchange_apply(h, scale)) return rc;
    COMM_NCCL(h, exchange_collective(2, (ncclComm_t)h->comm, h->d_xbucket, h->x_count, h->comm_world, h->comm_rank, h->f64, dl, h->stream));
    CMI_HIP(h, hipEventRecord(h->evx1, h->stream));
    h->exchange_timed = true;
    return CMI_OK;
}

// HIP-event time of the most recent cmi_comm_exchange on the instance's stream (pack .. loss all-reduce; includes the wait for the slowest rank)
extern "C" int cmi_comm_last_exchange_ms(cmi_handle h, float *ms) {
    if (!h || !ms) return CMI_E_INVALID;
    if (!h->exchange_timed) CMI_FAIL(h, CMI_E_INVALID, "comm_last_exchange_ms: no exchange has run");
    CMI_HIP(h, hipSetDevice(h->device));
    CMI_HIP(h, hipEventSynchronize(h->evx1));
    CMI_HIP(h, hipEventElapsedTime(ms, h->evx0, h->evx1));
    return CMI_OK;
}

// one global epoch of this rank: local epoch at `lrate`, the exchange with scale (1/W = the mean of the ranks' moves), the GLOBAL loss
// back -- the one host synchronisation of the epoch
extern "C" int cmi_comm_train_epoch(cmi_handle h, double lrate, double scale, double *global_loss) {
    if (!h) return CMI_E_INVALID;
    if (!h->comm) CMI_FAIL(h, CMI_E_INVALID, "comm_train_epoch: call cmi_comm_init first");
    if (int rc = cmi_train_epoch_async(h, lrate)) return rc;
    if (int rc = cmi_comm_exchange(h, scale)) return rc;
    double loss = 0.0;
    if (int rc = cmi_last_loss(h, &loss)) return rc;
    if (global_loss) *global_loss = loss;
    return CMI_OK;
}

// fm_api.cpp -- C ABI of the FM recommender (include/carskit_mi355x.h, cmi_fm_*).
#include "../../include/carskit_mi355x.h"
#include "env_knobs.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <numeric>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "fm_kernels.hpp"
#include "host_pool.hpp"
#include "rank_host.hpp"
#include "rank_kernels.hpp"

using namespace cmi;

// one field's stream on the device (fm_kernels.hpp FmOrder) and what owns it
struct FmOrderDev {
    FmRec *rec = nullptr;
    int32_t *piece_off = nullptr;
    int32_t count = 0;
    int64_t n_rec = 0;
};

// fields 0 / 1: the cell stream on the device (fm_kernels.hpp FmCells)
struct FmCellsDev {
    double *err0 = nullptr, *partial3 = nullptr, *w0part = nullptr;
    uint32_t *pk = nullptr;
    int32_t *fo = nullptr, *fcx = nullptr, *flag0 = nullptr, *bat_off = nullptr, *slot_off = nullptr, *slot_coord = nullptr, *cplx = nullptr;
    FmBatch *bat = nullptr;
    uint16_t *poff = nullptr;
    int32_t n_blocks = 0, n_slots = 0, n_cplx = 0, count = 0, S = 1, H = 1, n_batches = 0, n_flagged = 0;
    int64_t n_rec = 0, poff_len = 0, slice_len = 0;
};

struct cmi_fm_instance {
    int k = 0, n_users = 0, n_items = 0, n_conds = 0, n_ctx_dims = 1, device = 0;
    int64_t p = 0, n = 0, global_size = 0;
    std::string err;
    hipStream_t stream = nullptr;
    double *d_w0 = nullptr, *d_d0 = nullptr, *d_w = nullptr, *d_V = nullptr, *d_Vt = nullptr;
    bool v_valid = true, vt_valid = false; // which of V (p x k) / Vt (k x p) holds the current factors
    double *d_r = nullptr, *d_part = nullptr, *d_scratch = nullptr, *d_E = nullptr;
    double2 *d_tab = nullptr;
    int32_t *d_u = nullptr, *d_j = nullptr, *d_ctx = nullptr, *d_src[3] = {nullptr, nullptr, nullptr};
    FmOrderDev ord[3];   // [2] only: the context field (records sorted by feature, a wave per feature)
    FmCellsDev cell[2];  // users, items
    int atomic = 0;  // 0 (default): fm_cell_kernel, parking + a fixed walk, bit-reproducible; 1: CMI_FM_FLAG_RELAXED_SUMS, fm_cell_atomic_kernel (LDS atomics)
    int h_split = 0; // CMI_FM_HSPLIT: id-range parts per group (0 = chosen from the geometry)
    int batch_cap = FMC_RCAP, slot_cap = FMC_SLOTS; // experiment / test knobs (CMI_FM_BATCH, CMI_FM_SLOTS): smaller batches and blocks on small data
    RankWorkspace rank_ws; // cmi_fm_eval_rankings' buffers, reused by the next evaluation
    ncclComm_t comm = nullptr; // cmi_fm_comm_init: ratings sharded by user over one process per GPU
    int comm_world = 0;
    int col[3] = {-1, -1, -1}; // per field: the factor whose column sits in d_tab[].x of that field's coordinates (-1: none)
    int64_t part_count = 0;
    // experiment / test knob (CMI_FM_SLICE): an upper bound on the table entries of a slice of the gathered field; 0 = the geometry's own
    // choice (cells that fill a batch: fm_build_cells)
    int64_t slice_entries = 0;
    double regLw = 0, regLf = 0;
    bool have_ratings = false, have_model = false, initialised = false;
    int last_phase = -1;
};

static thread_local std::string g_fm_create_err;

#define FM_FAIL(h, code, ...)                                                                           \
    do {                                                                                                \
        char buf_[512];                                                                                 \
        snprintf(buf_, sizeof buf_, __VA_ARGS__);                                                       \
        (h)->err = buf_;                                                                                \
        return (code);                                                                                  \
    } while (0)
#define FM_HIP(h, expr)                                                                                 \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) FM_FAIL(h, CMI_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

extern "C" const char *cmi_fm_last_error(cmi_fm_handle h) { return h ? h->err.c_str() : g_fm_create_err.c_str(); }

static void fm_free_ratings(cmi_fm_instance *h) {
    void *ptrs[] = {h->d_r, h->d_u, h->d_j, h->d_ctx, h->d_src[0], h->d_src[1], h->d_src[2], h->d_E};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    h->d_r = h->d_E = nullptr;
    h->d_u = h->d_j = h->d_ctx = h->d_src[0] = h->d_src[1] = h->d_src[2] = nullptr;
    for (FmCellsDev &c : h->cell) {
        void *q[] = {c.err0, c.partial3, c.w0part, c.pk, c.fo, c.fcx, c.flag0, c.bat_off, c.slot_off, c.slot_coord, c.cplx, c.bat, c.poff};
        for (void *p : q)
            if (p) (void)hipFree(p);
        c = FmCellsDev();
    }
    for (FmOrderDev &o : h->ord) {
        void *q[] = {o.rec, o.piece_off};
        for (void *p : q)
            if (p) (void)hipFree(p);
        o = FmOrderDev();
    }
    h->have_ratings = h->initialised = false;
    h->n = 0;
}

extern "C" int cmi_fm_destroy(cmi_fm_handle h) {
    if (!h) return CMI_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    fm_free_ratings(h);
    if (h->comm) (void)ncclCommDestroy(h->comm);
    h->comm = nullptr;
    h->rank_ws.release();
    void *ptrs[] = {h->d_w0, h->d_d0, h->d_w, h->d_V, h->d_Vt, h->d_part, h->d_scratch, h->d_tab};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return CMI_OK;
}

extern "C" int cmi_fm_create(int k, int n_users, int n_items, int n_conds, int n_ctx_dims, int device,
                             unsigned flags, cmi_fm_handle *out) {
    if (out) *out = nullptr;
    if (!out || k <= 0 || n_users <= 0 || n_items <= 0 || n_conds < 0 || n_ctx_dims <= 0) {
        g_fm_create_err = "cmi_fm_create: invalid argument";
        return CMI_E_INVALID;
    }
    const int ndev = cmi_device_count();
    if (ndev <= 0) {
        g_fm_create_err = "cmi_fm_create: no HIP device visible (libcarskit_mi355x has no CPU fallback)";
        return CMI_E_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        g_fm_create_err = "cmi_fm_create: device index out of range";
        return CMI_E_INVALID;
    }
    cmi_fm_instance *h = new cmi_fm_instance();
    if (const char *v = getenv("CMI_FM_SLICE")) h->slice_entries = atoll(v); // experiment knob: 0 = one slice
    if (const char *v = getenv("CMI_FM_BATCH")) h->batch_cap = std::max(1, std::min(atoi(v), FMC_RCAP));
    if (const char *v = getenv("CMI_FM_SLOTS")) h->slot_cap = std::max((FMC_RCAP + FMC_RUN - 1) / FMC_RUN, std::min(atoi(v), FMC_SLOTS));
    // the reference's sweep is deterministic (FM.java:148-218): so is the default here; the relaxed (LDS-atomic) sums are an opt-in, and
    // an explicit CMI_FM_FLAG_DETERMINISTIC / CMI_FM_DETERMINISTIC=1 wins over the environment's opt-in
    h->atomic = ((flags & CMI_FM_FLAG_RELAXED_SUMS) || getenv("CMI_FM_RELAXED_SUMS")) && !(flags & CMI_FM_FLAG_DETERMINISTIC) && !getenv("CMI_FM_DETERMINISTIC") ? 1 : 0;
    if (const char *v = cmi_exp_env("CMI_FM_HSPLIT")) h->h_split = std::max(0, std::min(atoi(v), 8));
    h->k = k;
    h->n_users = n_users;
    h->n_items = n_items;
    h->n_conds = n_conds;
    h->n_ctx_dims = n_ctx_dims;
    h->device = device;
    h->p = (int64_t)n_users + n_items + n_conds;
    h->part_count = 2 * (int64_t)std::max(std::max(n_users, n_items), std::max(n_conds, 2));
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_w0, sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_d0, sizeof(double));
    if (e == hipSuccess) e = hipMemsetAsync(h->d_d0, 0, sizeof(double), h->stream);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_w, (size_t)h->p * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_V, (size_t)h->p * k * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_Vt, (size_t)h->p * k * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_part, (size_t)h->part_count * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_scratch, 256 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_tab, (size_t)(h->p + 1) * sizeof(double2)); // + 1: fm_rec_eval's dummy gather
    if (e == hipSuccess) e = hipMemsetAsync(h->d_tab, 0, (size_t)(h->p + 1) * sizeof(double2), h->stream);
    if (e != hipSuccess) {
        g_fm_create_err = std::string("cmi_fm_create: ") + hipGetErrorString(e);
        cmi_fm_destroy(h);
        return CMI_E_HIP;
    }
    *out = h;
    return CMI_OK;
}

extern "C" int cmi_fm_set_hparams(cmi_fm_handle h, double regLw, double regLf, int64_t global_size) {
    if (!h) return CMI_E_INVALID;
    h->regLw = regLw;
    h->regLf = regLf;
    h->global_size = global_size; // <= 0: use the local tuple count
    return CMI_OK;
}

extern "C" int cmi_fm_set_model(cmi_fm_handle h, double w0, const double *w, const double *V) {
    if (!h || !w || !V) return CMI_E_INVALID;
    FM_HIP(h, hipSetDevice(h->device));
    FM_HIP(h, hipMemcpyAsync(h->d_w0, &w0, sizeof(double), hipMemcpyHostToDevice, h->stream));
    FM_HIP(h, hipMemcpyAsync(h->d_w, w, (size_t)h->p * sizeof(double), hipMemcpyHostToDevice, h->stream));
    FM_HIP(h, hipMemcpyAsync(h->d_V, V, (size_t)h->p * h->k * sizeof(double), hipMemcpyHostToDevice, h->stream));
    FM_HIP(h, hipStreamSynchronize(h->stream));
    h->have_model = true;
    h->initialised = false;
    h->col[0] = h->col[1] = h->col[2] = -1;
    h->v_valid = true;
    h->vt_valid = false;
    return CMI_OK;
}

// the sweeps keep the factors column-major (Vt); everything else reads V
static int fm_sync_V(cmi_fm_instance *h) {
    if (h->v_valid) return CMI_OK;
    FM_HIP(h, fm_launch_transpose(h->d_Vt, h->d_V, h->k, h->p, h->stream));
    h->v_valid = true;
    return CMI_OK;
}
static int fm_sync_Vt(cmi_fm_instance *h) {
    if (h->vt_valid) return CMI_OK;
    FM_HIP(h, fm_launch_transpose(h->d_V, h->d_Vt, h->p, h->k, h->stream));
    h->vt_valid = true;
    return CMI_OK;
}

extern "C" int cmi_fm_get_model(cmi_fm_handle h, double *w0, double *w, double *V) {
    if (!h) return CMI_E_INVALID;
    FM_HIP(h, hipSetDevice(h->device));
    if (int rc = fm_sync_V(h)) return rc;
    if (w0) FM_HIP(h, hipMemcpyAsync(w0, h->d_w0, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (w) FM_HIP(h, hipMemcpyAsync(w, h->d_w, (size_t)h->p * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (V) FM_HIP(h, hipMemcpyAsync(V, h->d_V, (size_t)h->p * h->k * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    FM_HIP(h, hipStreamSynchronize(h->stream));
    return CMI_OK;
}

template <typename T>
static hipError_t up(T **dst, const std::vector<T> &v, hipStream_t s) {
    *dst = nullptr;
    if (v.empty()) return hipSuccess;
    hipError_t e = hipMalloc((void **)dst, v.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
}

// ---- the three streams (fm_kernels.hpp FmOrder), built once per cmi_fm_set_ratings ------------------------------------
struct FmOrderHost {
    std::vector<FmRec> rec;      // err0 is filled by cmi_fm_init on the device
    std::vector<int32_t> piece_off, src; // src[pos] = index of the rating in the caller's arrays
    int count = 0;
};

// The context field's stream: the ratings whose key[t] >= 0 (a context feature exists), sorted by key, stable in the caller's order;
// a record carries the rating's user (a) and item (c).
static void fm_build_order(int64_t n, const int32_t *key, const int32_t *a_id, const int32_t *c_id, int count, FmOrderHost &o) {
    o.count = count;
    o.piece_off.assign((size_t)count + 1, 0);
    for (int64_t t = 0; t < n; ++t)
        if (key[t] >= 0) o.piece_off[(size_t)key[t] + 1]++;
    for (int l = 0; l < count; ++l) o.piece_off[(size_t)l + 1] += o.piece_off[(size_t)l];
    const size_t in_support = (size_t)o.piece_off[(size_t)count];
    o.rec.resize(in_support);
    o.src.resize(in_support);
    std::vector<int32_t> cur(o.piece_off.begin(), o.piece_off.end() - 1);
    for (int64_t t = 0; t < n; ++t) {
        if (key[t] < 0) continue;
        const int32_t pos = cur[(size_t)key[t]]++;
        o.rec[pos] = FmRec{0.0, a_id[t], c_id[t]};
        o.src[pos] = (int32_t)t;
    }
}

// ---- the cell stream of fields 0 / 1 (fm_kernels.hpp FmCells), built once per cmi_fm_set_ratings -----------------------------------
struct FmCellsHost {
    std::vector<uint32_t> pk;
    std::vector<int32_t> fo, fcx, flag0, src, bat_off, slot_off, slot_coord, cplx;
    std::vector<FmBatch> bat;
    std::vector<uint16_t> poff;
    int n_blocks = 0, S = 1, H = 1, count = 0, n_flagged = 0;
    int64_t slice_len = 0;
};

// key[t]: this field's coordinate of rating t; other[t]: the id whose table entries the stream gathers (table entry other_base +
// other[t]); ctx[t]: the rating's context-combination id (a context feature exists iff ctx[t] < n_conds).
//
// Geometry.  GROUPS of consecutive coordinates with about the same number of records each and at most slot_cap accumulator slots (the
// slots live in the registers of the workgroup that walks the group).  How many lanes of a wave instruction share a line of the
// gathered table depends on the GROUP alone: a group holding n_G records puts 8 n_G / other_count of them on every 128-byte line, however
// the table is sliced -- so groups are as large as the slots allow, and when that leaves fewer groups than the chip has CUs, every
// group is walked by H workgroups, each taking the h-th part of every slice's id range (H id-range parts: more workgroups, the same
// sharing; the parts' sums meet in fm_cplx_kernel).  A BLOCK = (group, part) = one workgroup; it walks sub-slices h, h + H, h + 2H, ...
// of the gathered field; the records of (block, sub-slice) are a CELL, cut into batches of <= batch_cap records in gathered-id order
// (records with a context feature last).
static void fm_build_cells(int64_t n, const int32_t *key, const int32_t *other, const int32_t *ctx, int count, int other_count, int other_base,
                           int n_conds, int64_t slice_entries, int batch_cap, int slot_cap, int h_split, FmCellsHost &o, bool slot_in_word = false) {
    o.count = count;
    const bool times = getenv("CMI_SETUP_TIMES") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *w) {
        if (!times) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "fm cells (%d coordinates) %s %.3f s\n", count, w, std::chrono::duration<double>(t - T0).count());
        T0 = t;
    };
    // atomic form: the geometry (sub-slices, groups) is the deterministic form's, but a cell is cut into chunks of <= FMC_CHUNK records
    const int cut = slot_in_word ? std::min(batch_cap, FMC_CHUNK) : batch_cap;
    const int max_vs = (cut + FMC_RUN - 1) / FMC_RUN; // slots the longest possible run inside one batch needs
    std::vector<int32_t> deg((size_t)count, 0), vs((size_t)count, 1);
    parallel_ranges(n, host_threads(n), [&](int, int64_t b, int64_t e) { // (relaxed atomic increments: a count does not care about order)
        for (int64_t t = b; t < e; ++t) __atomic_fetch_add(&deg[(size_t)key[t]], 1, __ATOMIC_RELAXED);
    });
    // groups: first a bound on the slots (one per coordinate, more for coordinates hot enough to have runs > FMC_RUN inside a batch --
    // refined below once the sub-slices are known; a coordinate's records spread evenly over sub-slices only if the gathered ids do)
    int64_t ng_min = std::max<int64_t>(1, (int64_t)std::ceil((double)count / (0.97 * (double)slot_cap)));
    int64_t H = 1, NG = ng_min;
    if (h_split > 0) H = h_split;
    else if (ng_min < 256) {
        while (H < 8 && ng_min * H * 2 <= 256 && n / (ng_min * H * 2) >= 2048) H *= 2; // fill the CUs, but keep blocks worth a launch
    }
    if (NG * H > 256) NG = (NG * H + 255) / 256 * 256 / H; // whole waves of 256 workgroups
    else if (NG * H < 256 && h_split <= 0) NG = std::max<int64_t>(NG, std::min<int64_t>(256 / H, n / (2048 * H))); // small inputs: spread, blocks >= 2048 records
    NG = std::max<int64_t>(1, std::min<int64_t>(NG, count));
    const int64_t target = std::max<int64_t>(1, (n + NG - 1) / NG); // records per group
    // sub-slices: a cell should fill a batch: n / (NG * H * S) <= 0.93 * batch_cap; 17 bits of id per sub-slice at most
    int64_t S = std::max<int64_t>(1, (int64_t)std::ceil((double)n / ((double)NG * (double)H * 0.93 * (double)batch_cap)));
    if (slice_entries > 0) S = std::max<int64_t>(S, ((int64_t)other_count + slice_entries * H - 1) / (slice_entries * H)); // experiment knob: force smaller slices
    int64_t SS = S * H;
    while (((int64_t)other_count + SS - 1) / SS > 131072) SS += H;
    S = SS / H;
    const int64_t sub_len = std::max<int64_t>(1, ((int64_t)other_count + SS - 1) / SS);
    o.S = (int)S;
    o.H = (int)H;
    o.slice_len = sub_len;
    {   // slots per coordinate from its hottest cell (a run inside a batch is at most the cell's count)
        // (ratings with a context feature sit in one extra cell per block: counted per (coordinate, id-range part) in columns SS .. SS + H)
        const int64_t W = SS + H;
        std::vector<int32_t> cs((size_t)count * (size_t)W, 0);
        parallel_ranges(n, host_threads(n), [&](int, int64_t b, int64_t e) {
            for (int64_t t = b; t < e; ++t) {
                const int64_t ss = other[t] / sub_len;
                __atomic_fetch_add(&cs[(size_t)key[t] * (size_t)W + (size_t)(ctx[t] < n_conds ? SS + ss % H : ss)], 1, __ATOMIC_RELAXED);
            }
        });
        parallel_ranges(count, host_threads(count), [&](int, int64_t b, int64_t e) {
            for (int64_t l = b; l < e; ++l) {
                int32_t m = 0;
                for (int64_t s = 0; s < W; ++s) m = std::max(m, cs[(size_t)l * (size_t)W + (size_t)s]);
                vs[(size_t)l] = std::max(1, std::min(max_vs, (m + FMC_RUN - 1) / FMC_RUN));
            }
        });
    }
    lap("degrees + slots per coordinate");
    // a coordinate with more records than a group's target is cut by record rank into parts, each a group of its own
    std::vector<int32_t> grp_of((size_t)count, 0), loc((size_t)count, 0); // first group of a coordinate, its first slot inside the group
    std::vector<int32_t> grp_slots;
    {
        // a regular coordinate's group = where the running record count BEFORE it falls among NG equal shares: never more than NG groups
        // (one more would be a second wave of workgroups for one or two blocks) unless a group's slots overflow (then it is cut in two)
        // (the share weighs a coordinate by its records AND its slots, half each: ids are numbered in first-seen order, so late
        // coordinates have fewer records, and shares of records alone overflow the slots of the late groups)
        double n_reg = 0, s_reg = 0;
        for (int l = 0; l < count; ++l)
            if (!(deg[(size_t)l] > target && deg[(size_t)l] > 2 * (int64_t)batch_cap)) {
                n_reg += (double)deg[(size_t)l];
                s_reg += (double)vs[(size_t)l];
            }
        // the mix: as much weight on the records as keeps every share's slots within slot_cap (all weight on the slots always does)
        double alpha = 0.5;
        for (; alpha < 0.999; alpha += 0.1) {
            double c2 = 0;
            int64_t sh2 = 0, sl = 0, worst = 0;
            for (int l = 0; l < count; ++l) {
                if (deg[(size_t)l] > target && deg[(size_t)l] > 2 * (int64_t)batch_cap) continue;
                const int64_t sh = std::min<int64_t>(NG - 1, (int64_t)(c2 * (double)NG));
                if (sh != sh2) {
                    worst = std::max(worst, sl);
                    sl = 0;
                    sh2 = sh;
                }
                sl += vs[(size_t)l];
                c2 += (1.0 - alpha) * (n_reg > 0 ? (double)deg[(size_t)l] / n_reg : 0.0) + alpha * (double)vs[(size_t)l] / s_reg;
            }
            if (std::max(worst, sl) <= slot_cap) break;
        }
        alpha = std::min(alpha, 1.0);
        double cum = 0;
        int64_t share = -1;
        int32_t slots = 0;
        auto close = [&]() {
            grp_slots.push_back(slots);
            slots = 0;
        };
        for (int l = 0; l < count; ++l) {
            const int64_t d = deg[(size_t)l];
            if (d > target && d > 2 * (int64_t)batch_cap) { // giant: groups of its own
                if (slots > 0) close();
                grp_of[(size_t)l] = (int32_t)grp_slots.size();
                loc[(size_t)l] = 0;
                for (int64_t q = 0; q < (d + target - 1) / target; ++q) {
                    slots = vs[(size_t)l];
                    close();
                }
                continue;
            }
            const int64_t sh = std::min<int64_t>(NG - 1, (int64_t)(cum * (double)NG));
            if (slots > 0 && (sh != share || slots + vs[(size_t)l] > slot_cap)) close();
            share = sh;
            grp_of[(size_t)l] = (int32_t)grp_slots.size();
            loc[(size_t)l] = slots;
            slots += vs[(size_t)l];
            cum += (1.0 - alpha) * (n_reg > 0 ? (double)d / n_reg : 0.0) + alpha * (double)vs[(size_t)l] / s_reg;
        }
        if (slots > 0 || grp_slots.empty()) close();
    }
    auto giant = [&](int32_t l) { return deg[(size_t)l] > target && deg[(size_t)l] > 2 * (int64_t)batch_cap; };
    const int64_t G = (int64_t)grp_slots.size(), NB = G * H; // block = group * H + part
    o.n_blocks = (int)NB;
    o.slot_off.assign((size_t)NB + 1, 0);
    for (int64_t b = 0; b < NB; ++b) o.slot_off[(size_t)b + 1] = o.slot_off[(size_t)b] + grp_slots[(size_t)(b / H)];
    o.slot_coord.assign((size_t)o.slot_off[(size_t)NB], 0);
    for (int l = 0; l < count; ++l) {
        const int64_t parts = giant(l) ? (deg[(size_t)l] + target - 1) / target : 1;
        const int32_t v = vs[(size_t)l];
        const bool cplx = parts * H * v > 1;
        const int32_t first = o.slot_off[(size_t)((int64_t)grp_of[(size_t)l] * H)] + loc[(size_t)l];
        const int32_t stride = grp_slots[(size_t)grp_of[(size_t)l]]; // (a giant's groups hold only its v slots: stride = v)
        for (int64_t q = 0; q < parts * H; ++q)
            for (int32_t x = 0; x < v; ++x) o.slot_coord[(size_t)(first + q * stride + x)] = l | (cplx ? (int32_t)0x80000000 : 0);
        if (cplx) {
            o.cplx.push_back(l);
            o.cplx.push_back(first);
            o.cplx.push_back((int32_t)(parts * H) | (v << 16));
            o.cplx.push_back(stride);
        }
    }
    lap("groups");
    // every record's cell = (block, sub-slice index inside the block's walk), ratings with a context feature in an extra cell at the
    // block's end (index S); counting sort by cell, then gathered id inside the cell, stable in the caller's order
    const int64_t S1 = S + 1, NC = NB * S1;
    std::vector<int64_t> cell_off((size_t)NC + 1, 0);
    std::vector<int32_t> rcell((size_t)n);
    o.src.resize((size_t)n);
    bool any_giant = false;
    for (int l = 0; l < count && !any_giant; ++l) any_giant = giant(l);
    const int nt_cells = any_giant ? 1 : host_threads(n);
    if (nt_cells < 2) { // (a giant's records are dealt to its groups by their rank in the caller's order: one walk)
        std::vector<int32_t> rank((size_t)count, 0);
        for (int64_t t = 0; t < n; ++t) {
            const int32_t l = key[t];
            int64_t g = grp_of[(size_t)l];
            if (giant(l)) g += rank[(size_t)l]++ / target;
            const int64_t ss = other[t] / sub_len; // sub-slice: part ss % H, step ss / H of that part's walk
            const int64_t cell = (g * H + ss % H) * S1 + (ctx[t] < n_conds ? S : ss / H);
            rcell[(size_t)t] = (int32_t)cell;
            cell_off[(size_t)cell + 1]++;
        }
        for (int64_t c = 0; c < NC; ++c) cell_off[(size_t)c + 1] += cell_off[(size_t)c];
        std::vector<int64_t> cur(cell_off.begin(), cell_off.end() - 1);
        for (int64_t t = 0; t < n; ++t) o.src[(size_t)cur[(size_t)rcell[(size_t)t]]++] = (int32_t)t;
    } else { // the same stable counting sort in ranges: per-range histograms, offsets in range order, every range scatters its own records
        std::vector<std::vector<int64_t>> hist((size_t)nt_cells, std::vector<int64_t>((size_t)NC, 0));
        parallel_ranges(n, nt_cells, [&](int th, int64_t b, int64_t e) {
            std::vector<int64_t> &hh = hist[(size_t)th];
            for (int64_t t = b; t < e; ++t) {
                const int64_t g = grp_of[(size_t)key[t]];
                const int64_t ss = other[t] / sub_len;
                const int64_t cell = (g * H + ss % H) * S1 + (ctx[t] < n_conds ? S : ss / H);
                rcell[(size_t)t] = (int32_t)cell;
                hh[(size_t)cell]++;
            }
        });
        for (int64_t c = 0; c < NC; ++c) {
            int64_t run = cell_off[(size_t)c];
            for (int th = 0; th < nt_cells; ++th) {
                const int64_t cnt = hist[(size_t)th][(size_t)c];
                hist[(size_t)th][(size_t)c] = run; // first position of range th's records of cell c
                run += cnt;
            }
            cell_off[(size_t)c + 1] = run;
        }
        parallel_ranges(n, nt_cells, [&](int th, int64_t b, int64_t e) { // (the same ranges as above: parallel_ranges cuts [0, n) by nt alone)
            std::vector<int64_t> &cur = hist[(size_t)th];
            for (int64_t t = b; t < e; ++t) o.src[(size_t)cur[(size_t)rcell[(size_t)t]]++] = (int32_t)t;
        });
    }
    std::vector<int32_t>().swap(rcell);
    lap("cells: count + scatter");
    parallel_ranges(NC, host_threads(NC * 64), [&](int, int64_t cb, int64_t ce) {
        std::vector<uint64_t> keys; // {gathered id, rating}: one plain sort, no look-ups from the comparator
        for (int64_t c = cb; c < ce; ++c) {
            const int64_t c0 = cell_off[(size_t)c], len = cell_off[(size_t)c + 1] - c0;
            keys.resize((size_t)len);
            for (int64_t i = 0; i < len; ++i) {
                const int32_t t = o.src[(size_t)(c0 + i)];
                keys[(size_t)i] = ((uint64_t)(uint32_t)other[t] << 32) | (uint32_t)t;
            }
            std::sort(keys.begin(), keys.end());
            for (int64_t i = 0; i < len; ++i) o.src[(size_t)(c0 + i)] = (int32_t)(uint32_t)keys[(size_t)i];
        }
    });
    lap("sort inside cells");
    // batches (cells cut into <= batch_cap records), their slot boundaries and every record's parked position
    o.bat_off.assign((size_t)NB + 1, 0);
    std::vector<int64_t> bat_first((size_t)NB + 1, 0), poff_first((size_t)NB + 1, 0);
    for (int64_t b = 0; b < NB; ++b) {
        int64_t nbat = 0;
        for (int64_t s = 0; s < S1; ++s) {
            const int64_t len = cell_off[(size_t)(b * S1 + s) + 1] - cell_off[(size_t)(b * S1 + s)];
            nbat += (len + cut - 1) / cut;
        }
        bat_first[(size_t)b + 1] = bat_first[(size_t)b] + nbat;
        poff_first[(size_t)b + 1] = poff_first[(size_t)b] + (slot_in_word ? 0 : nbat * ((int64_t)grp_slots[(size_t)(b / H)] + 1)); // no slot boundaries in the atomic form
        o.bat_off[(size_t)b + 1] = (int32_t)bat_first[(size_t)b + 1];
    }
    o.bat.resize((size_t)bat_first[(size_t)NB]);
    o.poff.assign((size_t)poff_first[(size_t)NB], 0);
    o.pk.resize((size_t)n);
    o.flag0.assign((size_t)NB, 0);
    // the side arrays of the ratings with a context feature: a block's entries are contiguous, in stream order
    std::vector<int64_t> flag_first((size_t)NB + 1, 0);
    for (int64_t b = 0; b < NB; ++b)
        flag_first[(size_t)b + 1] = flag_first[(size_t)b] + (cell_off[(size_t)(b * S1 + S) + 1] - cell_off[(size_t)(b * S1 + S)]);
    o.fo.resize((size_t)flag_first[(size_t)NB]);
    o.fcx.resize((size_t)flag_first[(size_t)NB]);
    o.n_flagged = (int)flag_first[(size_t)NB];
    parallel_ranges(NB, host_threads(NB * 4096), [&](int, int64_t bb, int64_t be) {
        std::vector<int32_t> run, cnt, cursor;
        for (int64_t b = bb; b < be; ++b) {
            const int32_t ns = grp_slots[(size_t)(b / H)];
            const int64_t h = b % H;
            int64_t bi = bat_first[(size_t)b], pf = poff_first[(size_t)b];
            cnt.assign((size_t)ns + 1, 0);
            cursor.assign((size_t)ns, 0);
            int64_t ff = flag_first[(size_t)b];
            for (int64_t s = 0; s < S1; ++s) {
                const int64_t c0 = cell_off[(size_t)(b * S1 + s)], c1 = cell_off[(size_t)(b * S1 + s) + 1];
                const bool fl = s == S;                              // the block's ratings with a context feature
                const int64_t id0 = fl ? 0 : (s * H + h) * sub_len; // first gathered id of the cell's range
                if (fl) o.flag0[(size_t)b] = (int32_t)bi;
                for (int64_t r0 = c0; r0 < c1; r0 += cut) {
                    const int64_t r1 = std::min<int64_t>(c1, r0 + cut);
                    // slot of a record: the coordinate's first slot in the group + (its rank inside the batch's run) / FMC_RUN
                    if (slot_in_word) { // atomic form: no positions, no boundaries -- work proportional to the chunk, not to the group's slots
                        if (run.size() < (size_t)ns) run.assign((size_t)ns, 0); // (all zero between chunks)
                        const int64_t ff0 = ff;
                        for (int64_t r = r0; r < r1; ++r) {
                            const int32_t t = o.src[(size_t)r], first = loc[(size_t)key[t]];
                            const uint32_t the_slot = (uint32_t)(first + run[(size_t)first]++ / FMC_RUN);
                            o.pk[(size_t)r] = (fl ? 0u : (uint32_t)(other[t] - id0)) | (the_slot << 17);
                            if (fl) {
                                o.fo[(size_t)ff] = other[t];
                                o.fcx[(size_t)ff++] = ctx[t];
                            }
                        }
                        for (int64_t r = r0; r < r1; ++r) run[(size_t)loc[(size_t)key[o.src[(size_t)r]]]] = 0;
                        o.bat[(size_t)bi++] = FmBatch{(int32_t)r0, (int32_t)(r1 - r0), fl ? (int32_t)ff0 : (int32_t)(other_base + id0), (int32_t)pf,
                                                      fl ? (int32_t)(r1 - r0) : 0, 0};
                        continue;
                    }
                    run.assign((size_t)ns, 0); // per FIRST slot of a coordinate: its records seen so far in this batch
                    std::fill(cnt.begin(), cnt.end(), 0);
                    for (int64_t r = r0; r < r1; ++r) {
                        const int32_t first = loc[(size_t)key[o.src[(size_t)r]]];
                        cnt[(size_t)(first + run[(size_t)first]++ / FMC_RUN) + 1]++;
                    }
                    for (int32_t q = 0; q < ns; ++q) cnt[(size_t)q + 1] += cnt[(size_t)q];
                    if (!slot_in_word) {
                        uint16_t *po = o.poff.data() + pf;
                        for (int32_t q = 0; q <= ns; ++q) po[q] = (uint16_t)cnt[(size_t)q];
                    }
                    std::fill(run.begin(), run.end(), 0);
                    std::copy(cnt.begin(), cnt.end() - 1, cursor.begin());
                    const int64_t ff0 = ff;
                    for (int64_t r = r0; r < r1; ++r) {
                        const int32_t t = o.src[(size_t)r], first = loc[(size_t)key[t]];
                        const int32_t the_slot = first + run[(size_t)first]++ / FMC_RUN;
                        const uint32_t pos = slot_in_word ? (uint32_t)the_slot : (uint32_t)cursor[(size_t)the_slot]++;
                        o.pk[(size_t)r] = (fl ? 0u : (uint32_t)(other[t] - id0)) | (pos << 17);
                        if (fl) {
                            o.fo[(size_t)ff] = other[t];
                            o.fcx[(size_t)ff++] = ctx[t];
                        }
                    }
                    o.bat[(size_t)bi++] = FmBatch{(int32_t)r0, (int32_t)(r1 - r0), fl ? (int32_t)ff0 : (int32_t)(other_base + id0), (int32_t)pf,
                                                  fl ? (int32_t)(r1 - r0) : 0, 0};
                    if (!slot_in_word) pf += (int64_t)ns + 1;
                }
            }
        }
    });
    lap("batches + packed words");
    if (getenv("CMI_FM_DEBUG")) {
        int64_t maxcell = 0, maxblk = 0, minblk = n;
        for (int64_t c = 0; c < NC; ++c) maxcell = std::max(maxcell, cell_off[(size_t)c + 1] - cell_off[(size_t)c]);
        for (int64_t b = 0; b < NB; ++b) {
            const int64_t r = cell_off[(size_t)((b + 1) * S1)] - cell_off[(size_t)(b * S1)];
            maxblk = std::max(maxblk, r);
            minblk = std::min(minblk, r);
        }
        int32_t maxslots = 0;
        for (int32_t x : grp_slots) maxslots = std::max(maxslots, x);
        fprintf(stderr, "[cmi fm] cells: count %d other %d n %lld: planned NG %lld H %lld S %lld sub_len %lld target %lld -> groups %lld blocks %lld batches %zu "
                "max cell %lld block records %lld..%lld max slots %d complex %zu\n", count, other_count, (long long)n, (long long)NG, (long long)H, (long long)S,
                (long long)sub_len, (long long)target, (long long)G, (long long)NB, o.bat.size(), (long long)maxcell, (long long)minblk, (long long)maxblk, maxslots,
                o.cplx.size() / 4);
    }
}

static hipError_t fm_upload_cells(const FmCellsHost &o, FmCellsDev &d, hipStream_t s) {
    d.n_blocks = o.n_blocks;
    d.n_slots = (int32_t)o.slot_coord.size();
    d.n_cplx = (int32_t)(o.cplx.size() / 4);
    d.H = o.H;
    d.count = o.count;
    d.S = o.S;
    d.n_batches = (int32_t)o.bat.size();
    d.n_flagged = o.n_flagged;
    d.n_rec = (int64_t)o.pk.size();
    d.poff_len = (int64_t)o.poff.size();
    d.slice_len = o.slice_len;
    hipError_t e = up(&d.pk, o.pk, s);
    if (e == hipSuccess) e = up(&d.fo, o.fo, s);
    if (e == hipSuccess) e = up(&d.fcx, o.fcx, s);
    if (e == hipSuccess) e = up(&d.flag0, o.flag0, s);
    if (e == hipSuccess) e = up(&d.bat, o.bat, s);
    if (e == hipSuccess) e = up(&d.bat_off, o.bat_off, s);
    if (e == hipSuccess) e = up(&d.poff, o.poff, s);
    if (e == hipSuccess) e = up(&d.slot_off, o.slot_off, s);
    if (e == hipSuccess) e = up(&d.slot_coord, o.slot_coord, s);
    if (e == hipSuccess) e = up(&d.cplx, o.cplx, s);
    if (e == hipSuccess && d.n_rec > 0) e = hipMalloc((void **)&d.err0, (size_t)d.n_rec * sizeof(double));
    if (e == hipSuccess && d.n_rec > 0) e = hipMemsetAsync(d.err0, 0, (size_t)d.n_rec * sizeof(double), s);
    if (e == hipSuccess && d.n_slots > 0) e = hipMalloc((void **)&d.partial3, (size_t)d.n_slots * 3 * sizeof(double));
    if (e == hipSuccess && d.n_slots > 0) e = hipMalloc((void **)&d.w0part, (size_t)d.n_slots * sizeof(double));
    return e;
}

static hipError_t fm_upload_order(const FmOrderHost &o, FmOrderDev &d, hipStream_t s) {
    d.count = o.count;
    d.n_rec = (int64_t)o.rec.size();
    hipError_t e = up(&d.rec, o.rec, s);
    if (e == hipSuccess) e = up(&d.piece_off, o.piece_off, s);
    return e;
}

static int fm_set_ratings_impl(cmi_fm_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, const double *r);

// The exception barrier of the boundary: the cell streams allocate O(tuples) host memory, part of it on the host pool's
// threads (host_pool.hpp hands a range body's exception to the caller) and on the two side threads below; nothing C++ may cross into a
// C / JNI / ctypes host.
extern "C" int cmi_fm_set_ratings(cmi_fm_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx,
                                  const double *r) {
    if (!h) return CMI_E_INVALID;
    try {
        return fm_set_ratings_impl(h, n, u, j, ctx, r);
    } catch (const std::exception &e) {
        fm_free_ratings(h);
        FM_FAIL(h, CMI_E_HOST, "fm_set_ratings: host-side failure: %s", e.what());
    } catch (...) {
        fm_free_ratings(h);
        FM_FAIL(h, CMI_E_HOST, "fm_set_ratings: host-side failure (unknown exception)");
    }
}

static int fm_set_ratings_impl(cmi_fm_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx, const double *r) {
    if (n < 0 || (n > 0 && (!u || !j || !ctx || !r))) FM_FAIL(h, CMI_E_INVALID, "fm_set_ratings: null arrays");
    if (n >= ((int64_t)1 << 31)) FM_FAIL(h, CMI_E_UNSUPPORTED, "fm_set_ratings: more than 2^31-1 tuples");
    for (int64_t t = 0; t < n; ++t)
        if (u[t] < 0 || u[t] >= h->n_users || j[t] < 0 || j[t] >= h->n_items || ctx[t] < 0)
            FM_FAIL(h, CMI_E_INVALID, "fm_set_ratings: id out of range at tuple %lld", (long long)t);
    FM_HIP(h, hipSetDevice(h->device));
    FM_HIP(h, hipStreamSynchronize(h->stream));
    fm_free_ratings(h);
    const bool times = getenv("CMI_SETUP_TIMES") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *w) {
        if (!times) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "fm_set_ratings %s %.3f s\n", w, std::chrono::duration<double>(t - T0).count());
        T0 = t;
    };
    lap("validate");
    FmCellsHost cu, ci;
    FmOrderHost oc;
    {
        // the three streams do not depend on each other: the item cells and the context order are built on threads of their own beside
        // the user cells
        std::vector<int32_t> ckey((size_t)n);
        auto item_cells = [&]() {
            fm_build_cells(n, j, u, ctx, h->n_items, h->n_users, 0, h->n_conds, h->slice_entries, h->batch_cap, h->slot_cap, h->h_split, ci, h->atomic != 0);
        };
        auto ctx_order = [&]() {
            // context features: only ratings whose context-combination id is < numConditions have one (FM.java:81-86)
            for (int64_t t = 0; t < n; ++t) ckey[(size_t)t] = ctx[t] < h->n_conds ? ctx[t] : -1;
            fm_build_order(n, ckey.data(), u, j, h->n_conds, oc);
        };
        // a side thread's exception (std::bad_alloc ...) is caught IN the thread and rethrown here after both have been joined: an
        // exception escaping a std::thread body, or unwinding past a joinable std::thread, would be std::terminate
        std::exception_ptr xi, xc, xu;
        auto guarded = [](auto &body, std::exception_ptr &x) {
            return [&body, &x]() {
                try {
                    body();
                } catch (...) {
                    x = std::current_exception();
                }
            };
        };
        std::thread ti, tc;
        bool hi = true, hc = true;
        try {
            ti = std::thread(guarded(item_cells, xi));
        } catch (const std::system_error &) { // the process may not create more threads: one after the other
            hi = false;
        }
        try {
            tc = std::thread(guarded(ctx_order, xc));
        } catch (const std::system_error &) {
            hc = false;
        }
        try {
            fm_build_cells(n, u, j, ctx, h->n_users, h->n_items, h->n_users, h->n_conds, h->slice_entries, h->batch_cap, h->slot_cap, h->h_split, cu, h->atomic != 0);
        } catch (...) {
            xu = std::current_exception();
        }
        if (hi) ti.join();
        if (hc) tc.join();
        if (xu) std::rethrow_exception(xu);
        if (xi) std::rethrow_exception(xi);
        if (xc) std::rethrow_exception(xc);
        if (!hi) item_cells();
        if (!hc) ctx_order();
    }
    lap("cells + context order (three threads)");
    // the ratings as plain arrays in the caller's order (cmi_fm_init computes err0 there; every stream copies its err0 through `src`)
    hipError_t e = hipSuccess;
    {
        auto up_raw = [&](auto **dst, const auto *src) { // straight from the caller's arrays (they outlive the synchronize below)
            *dst = nullptr;
            if (n == 0) return hipSuccess;
            hipError_t e2 = hipMalloc((void **)dst, (size_t)n * sizeof(**dst));
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(*dst, src, (size_t)n * sizeof(**dst), hipMemcpyHostToDevice, h->stream);
            return e2;
        };
        e = up_raw(&h->d_u, u);
        if (e == hipSuccess) e = up_raw(&h->d_j, j);
        if (e == hipSuccess) e = up_raw(&h->d_ctx, ctx);
        if (e == hipSuccess) e = up_raw(&h->d_r, r);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    lap("tuple upload");
    if (e == hipSuccess && n > 0) e = hipMalloc((void **)&h->d_E, (size_t)n * sizeof(double));
    if (e == hipSuccess) e = up(&h->d_src[0], cu.src, h->stream);
    if (e == hipSuccess) e = up(&h->d_src[1], ci.src, h->stream);
    if (e == hipSuccess) e = up(&h->d_src[2], oc.src, h->stream);
    if (e == hipSuccess) e = fm_upload_cells(cu, h->cell[0], h->stream);
    if (e == hipSuccess) e = fm_upload_cells(ci, h->cell[1], h->stream);
    if (e == hipSuccess) e = fm_upload_order(oc, h->ord[2], h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        fm_free_ratings(h);
        FM_FAIL(h, CMI_E_HIP, "fm_set_ratings: upload failed: %s", hipGetErrorString(e));
    }
    lap("stream upload");
    h->n = n;
    h->have_ratings = true;
    return CMI_OK;
}

static FmArgs fm_args(cmi_fm_instance *h) {
    FmArgs a;
    a.w0 = h->d_w0;
    a.d0 = h->d_d0;
    a.w = h->d_w;
    a.V = h->d_V;
    a.Vt = h->d_Vt;
    a.tab = h->d_tab;
    for (int f = 0; f < 3; ++f) {
        const FmOrderDev &d = h->ord[f];
        a.ord[f] = FmOrder{d.rec, d.piece_off, d.count, d.n_rec};
    }
    a.ord[0].count = h->n_users; // (fields 0 / 1 stream cells; the apply kernel reads `count`)
    a.ord[1].count = h->n_items;
    for (int f = 0; f < 2; ++f) {
        const FmCellsDev &c = h->cell[f];
        a.cell[f] = FmCells{c.err0, c.pk, c.fo, c.fcx, c.bat, c.bat_off, c.flag0, c.poff, c.slot_off, c.slot_coord, c.partial3, c.w0part, c.cplx,
                            c.n_blocks, c.n_slots, c.n_cplx, c.count, c.S, c.n_rec};
    }
    a.part = h->d_part;
    a.u = h->d_u;
    a.j = h->d_j;
    a.ctx = h->d_ctx;
    a.r = h->d_r;
    a.E = h->d_E;
    for (int f = 0; f < 3; ++f) a.src[f] = h->d_src[f];
    a.n = h->n;
    a.global_size = h->global_size > 0 ? h->global_size : h->n;
    a.k = h->k;
    a.n_users = h->n_users;
    a.n_items = h->n_items;
    a.n_conds = h->n_conds;
    a.xcol = -1;
    a.atomic = h->atomic;
    a.xc = 1.0 / (double)h->n_ctx_dims;
    a.regLw = h->regLw;
    a.regLf = h->regLf;
    return a;
}

static int fm_ready(cmi_fm_instance *h, bool need_init) {
    if (!h->have_ratings) FM_FAIL(h, CMI_E_INVALID, "fm: call cmi_fm_set_ratings first");
    if (!h->have_model) FM_FAIL(h, CMI_E_INVALID, "fm: call cmi_fm_set_model first");
    if (need_init && !h->initialised) FM_FAIL(h, CMI_E_INVALID, "fm: call cmi_fm_init first");
    FM_HIP(h, hipSetDevice(h->device));
    return CMI_OK;
}

extern "C" int cmi_fm_init(cmi_fm_handle h) {
    if (!h) return CMI_E_INVALID;
    if (int rc = fm_ready(h, false)) return rc;
    if (int rc = fm_sync_V(h)) return rc;
    h->col[0] = h->col[1] = h->col[2] = -1;
    FM_HIP(h, fm_launch_init(fm_args(h), h->stream));
    FM_HIP(h, hipStreamSynchronize(h->stream));
    h->initialised = true;
    return CMI_OK;
}

// phase numbering: 0 = w0; 1,2,3 = w of users/items/context features; 4 + 3*f + field = column f of V
extern "C" int cmi_fm_num_phases(cmi_fm_handle h) { return h ? 4 + 3 * h->k : 0; }

static bool phase_decode(cmi_fm_instance *h, int phase, int *field, int *f) {
    if (phase < 0 || phase >= 4 + 3 * h->k) return false;
    if (phase == 0) {
        *field = -1;
        *f = -1;
    } else if (phase < 4) {
        *field = phase - 1;
        *f = -1;
    } else {
        *field = (phase - 4) % 3;
        *f = (phase - 4) / 3;
    }
    return true;
}

// ---- phase driver.  Errors are never stored (fm_kernels.hip header), so the phases may be driven in any order; the
// only state between them is which column of V sits in tab[].x.
// the column an update of (field, f) leaves in tab[].x of the coordinates it updates (fm_kernels.hip fm_update)
static int fm_xcol(const cmi_fm_instance *h, int field, int f) {
    if (field == 0) return f;                // users: the column just written (-1: a linear-weight phase leaves .x alone)
    return std::min(f + 1, h->k - 1);        // items / context features: the next factor's column
}

// What a phase's GATHERS need in tab[].x: the user phase of factor f gathers the items' (and the context features') column f, the item
// phase the users' (and the context features'); the context phase reads its columns out of Vt.  In the order of a sweep the updates have
// left exactly these columns (fm_update); phases driven in another order reload what is missing.
static int fm_before_phase(cmi_fm_instance *h, int field, int f) {
    if (int rc = fm_sync_Vt(h)) return rc;
    if (f < 0 || field == 2) return CMI_OK;
    for (int g = 0; g < 3; ++g) {
        if (g == field || h->col[g] == f) continue;
        FM_HIP(h, fm_launch_col_load(fm_args(h), g, f, h->stream));
        h->col[g] = f;
    }
    return CMI_OK;
}
static void fm_after_update(cmi_fm_instance *h, int field, int f) {
    const int x = fm_xcol(h, field, f);
    if (x >= 0) h->col[field] = x;
    if (f >= 0) h->v_valid = false;
}

extern "C" int cmi_fm_phase_reduce(cmi_fm_handle h, int phase) {
    if (!h) return CMI_E_INVALID;
    if (int rc = fm_ready(h, true)) return rc;
    int field, f;
    if (!phase_decode(h, phase, &field, &f)) FM_FAIL(h, CMI_E_INVALID, "fm: bad phase %d", phase);
    if (int rc = fm_before_phase(h, field, f)) return rc;
    const FmArgs a = fm_args(h);
    if (phase == 0) {
        FM_HIP(h, fm_launch_w0_reduce(a, h->d_scratch, h->stream));
    } else {
        FM_HIP(h, fm_launch_phase(a, field, f, 0, h->stream));
    }
    h->last_phase = phase;
    return CMI_OK;
}

extern "C" int cmi_fm_phase_buffer(cmi_fm_handle h, int phase, void **dev_ptr, int64_t *count) {
    if (!h || !dev_ptr || !count) return CMI_E_INVALID;
    int field, f;
    if (!phase_decode(h, phase, &field, &f)) FM_FAIL(h, CMI_E_INVALID, "fm: bad phase %d", phase);
    *dev_ptr = h->d_part;
    *count = phase == 0 ? 2 : 2 * (int64_t)(field == 0 ? h->n_users : field == 1 ? h->n_items : h->n_conds);
    return CMI_OK;
}

extern "C" int cmi_fm_phase_apply(cmi_fm_handle h, int phase) {
    if (!h) return CMI_E_INVALID;
    if (int rc = fm_ready(h, true)) return rc;
    int field, f;
    if (!phase_decode(h, phase, &field, &f)) FM_FAIL(h, CMI_E_INVALID, "fm: bad phase %d", phase);
    if (h->last_phase != phase) FM_FAIL(h, CMI_E_INVALID, "fm: phase_apply(%d) without the matching phase_reduce", phase);
    FmArgs a = fm_args(h);
    if (phase == 0) FM_HIP(h, fm_launch_w0_apply(a, h->stream));
    else {
        a.xcol = fm_xcol(h, field, f);
        FM_HIP(h, fm_launch_apply(a, field, f, h->stream));
        fm_after_update(h, field, f);
    }
    h->last_phase = -1;
    return CMI_OK;
}

// one phase, reduce + update with no exchange point: what cmi_fm_sweep runs per phase; a multi-GPU host uses it for the
// phases whose coordinates live on one rank only (the user field of user-sharded ratings)
extern "C" int cmi_fm_phase_run(cmi_fm_handle h, int phase) {
    if (!h) return CMI_E_INVALID;
    if (int rc = fm_ready(h, true)) return rc;
    int field, f;
    if (!phase_decode(h, phase, &field, &f)) FM_FAIL(h, CMI_E_INVALID, "fm: bad phase %d", phase);
    if (int rc = fm_before_phase(h, field, f)) return rc;
    FmArgs a = fm_args(h);
    if (phase == 0) {
        FM_HIP(h, fm_launch_w0_reduce(a, h->d_scratch, h->stream));
        FM_HIP(h, fm_launch_w0_apply(a, h->stream));
    } else {
        a.xcol = fm_xcol(h, field, f);
        FM_HIP(h, fm_launch_phase(a, field, f, 2, h->stream));
        fm_after_update(h, field, f);
    }
    h->last_phase = -1;
    return CMI_OK;
}

extern "C" int cmi_fm_sweep(cmi_fm_handle h) {
    if (!h) return CMI_E_INVALID;
    const int np = cmi_fm_num_phases(h);
    for (int ph = 0; ph < np; ++ph)
        if (int rc = cmi_fm_phase_run(h, ph)) return rc;
    return CMI_OK;
}


// ---- measurement helpers (bench.py) ---------------------------------------------------------------------------------
// out[0..1] slices of the user / item order, [2..4] records of the three orders, [5..6] chunks of the user / item order,
// [7] HBM bytes one factor (3 phases + the column load) has to move with this layout, [8] the same for the reduce launch of
// the user field alone, [9] of the item field alone, [10] slice entries, [11] p.
extern "C" int cmi_fm_layout(cmi_fm_handle h, int64_t out[12]) {
    if (!h || !out) return CMI_E_INVALID;
    if (!h->have_ratings) FM_FAIL(h, CMI_E_INVALID, "fm: call cmi_fm_set_ratings first");
    int64_t factor = 0, red[3] = {0, 0, 0};
    for (int f = 0; f < 2; ++f) {
        const FmCellsDev &c = h->cell[f];
        // the launch of a user / item phase: the records (8-byte error + 4-byte packed word), a 64-byte sector per record with a context
        // feature (its combination id), the batches' slot boundaries and descriptors, the slots' coordinates, every coordinate's table
        // entry read and written and its Vt entry written (the update is part of the launch), complex coordinates' sums out and back in;
        // the gathered table entries are L2-resident by construction and are charged once per XCD and slice
        red[f] = c.n_rec * 12 + (int64_t)c.n_flagged * 8 + (h->atomic ? 0 : c.poff_len * 2) + (int64_t)c.n_batches * 16 + (int64_t)c.n_blocks * 16 +
                 (int64_t)c.n_slots * 4 + (int64_t)c.count * (16 + 16 + 8) + (c.n_cplx > 0 ? (int64_t)c.n_slots * 2 * 24 + (int64_t)c.n_cplx * 16 : 0);
        const int64_t other = f == 0 ? h->n_items : h->n_users;
        red[f] += 8 * other * 16;
        factor += red[f];
    }
    {
        const FmOrderDev &o = h->ord[2];
        red[2] = o.n_rec * (16 + 2 * 16 + 2 * 8) + ((int64_t)o.count + 1) * 4 + (int64_t)o.count * (16 + 16 + 8 + 8); // records + their gathers, entries
        factor += red[2];
    }
    out[0] = h->cell[0].S;
    out[1] = h->cell[1].S;
    out[2] = h->cell[0].n_rec;
    out[3] = h->cell[1].n_rec;
    out[4] = h->ord[2].n_rec;
    out[5] = h->cell[0].n_batches;
    out[6] = h->cell[1].n_batches;
    out[7] = factor;
    out[8] = red[0];
    out[9] = red[1];
    out[10] = h->cell[0].slice_len * h->cell[0].H;
    out[11] = h->p;
    return CMI_OK;
}

// Average duration (HIP events on the instance's stream) of `reps` launches of the REDUCE kernel of one phase -- it only
// writes the partial-sum scratch, so the model is untouched.
extern "C" int cmi_fm_time_reduce(cmi_fm_handle h, int phase, int reps, double *avg_ms) {
    if (!h || !avg_ms || reps < 1) return CMI_E_INVALID;
    if (int rc = fm_ready(h, true)) return rc;
    int field, f;
    if (!phase_decode(h, phase, &field, &f) || phase == 0) FM_FAIL(h, CMI_E_INVALID, "fm: bad phase %d", phase);
    if (int rc = fm_before_phase(h, field, f)) return rc;
    const FmArgs a = fm_args(h);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    FM_HIP(h, hipEventCreate(&e0));
    FM_HIP(h, hipEventCreate(&e1));
    hipError_t e = fm_launch_reduce_only(a, field, f, h->stream); // warm
    if (e == hipSuccess) e = hipEventRecord(e0, h->stream);
    for (int i = 0; i < reps && e == hipSuccess; ++i) e = fm_launch_reduce_only(a, field, f, h->stream);
    if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    FM_HIP(h, e);
    *avg_ms = (double)ms / reps;
    return CMI_OK;
}


// ---- ratings sharded by user over one process per GPU (BASELINE configs[3]): the sweep with its per-phase exchange issued by the
// library itself.  User phases are rank-local (a user's ratings live on one rank) and run fused; the w0 / item / context phases
// all-reduce their [num | den] buffer between reduce and apply on the instance's stream -- ~130 small collectives per sweep with no
// host code between them (driven from Python they cost more than the phases).  Every rank then applies the same update, so the
// replicated item / context part of the model stays identical.  cmi_fm_set_hparams' global_size must be the ratings of ALL ranks.
extern "C" int cmi_fm_comm_init(cmi_fm_handle h, const void *id, int rank, int world) {
    if (!h || !id || world < 1 || rank < 0 || rank >= world) return CMI_E_INVALID;
    static_assert(sizeof(ncclUniqueId) == CMI_COMM_ID_BYTES, "CMI_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
    FM_HIP(h, hipSetDevice(h->device));
    if (h->comm) (void)ncclCommDestroy(h->comm);
    h->comm = nullptr;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    const ncclResult_t r = ncclCommInitRank(&h->comm, world, u, rank);
    if (r != ncclSuccess) FM_FAIL(h, CMI_E_HIP, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    h->comm_world = world;
    return CMI_OK;
}

extern "C" int cmi_fm_comm_sweep(cmi_fm_handle h) {
    if (!h) return CMI_E_INVALID;
    if (!h->comm) FM_FAIL(h, CMI_E_INVALID, "fm_comm_sweep: call cmi_fm_comm_init first");
    const int np = cmi_fm_num_phases(h);
    for (int ph = 0; ph < np; ++ph) {
        int field, f;
        phase_decode(h, ph, &field, &f);
        if (field == 0) { // users: local to this rank
            if (int rc = cmi_fm_phase_run(h, ph)) return rc;
            continue;
        }
        if (int rc = cmi_fm_phase_reduce(h, ph)) return rc;
        void *buf = nullptr;
        int64_t cnt = 0;
        if (int rc = cmi_fm_phase_buffer(h, ph, &buf, &cnt)) return rc;
        const ncclResult_t r = ncclAllReduce(buf, buf, (size_t)cnt, ncclDouble, ncclSum, h->comm, h->stream);
        if (r != ncclSuccess) FM_FAIL(h, CMI_E_HIP, "ncclAllReduce (phase %d) failed: %s", ph, ncclGetErrorString(r));
        if (int rc = cmi_fm_phase_apply(h, ph)) return rc;
    }
    return CMI_OK;
}

extern "C" int cmi_fm_train(cmi_fm_handle h, int num_iters) {
    if (!h) return CMI_E_INVALID;
    if (int rc = cmi_fm_init(h)) return rc;
    for (int it = 0; it < num_iters; ++it)
        if (int rc = cmi_fm_sweep(h)) return rc;
    FM_HIP(h, hipStreamSynchronize(h->stream));
    return CMI_OK;
}

extern "C" int cmi_fm_stream(cmi_fm_handle h, void **stream) {
    if (!h || !stream) return CMI_E_INVALID;
    *stream = (void *)h->stream;
    return CMI_OK;
}

extern "C" int cmi_fm_synchronize(cmi_fm_handle h) {
    if (!h) return CMI_E_INVALID;
    FM_HIP(h, hipSetDevice(h->device));
    FM_HIP(h, hipStreamSynchronize(h->stream));
    return CMI_OK;
}

extern "C" int cmi_fm_predict_batch(cmi_fm_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx,
                                    int bound, double lo, double hi, double *out) {
    if (!h) return CMI_E_INVALID;
    if (!h->have_model) FM_FAIL(h, CMI_E_INVALID, "fm: call cmi_fm_set_model first");
    if (n < 0 || (n > 0 && (!u || !j || !ctx || !out))) FM_FAIL(h, CMI_E_INVALID, "fm_predict: null arrays");
    for (int64_t t = 0; t < n; ++t)
        if (u[t] < 0 || u[t] >= h->n_users || j[t] < 0 || j[t] >= h->n_items || ctx[t] < 0)
            FM_FAIL(h, CMI_E_INVALID, "fm_predict: id out of range at tuple %lld", (long long)t);
    if (n == 0) return CMI_OK;
    FM_HIP(h, hipSetDevice(h->device));
    if (int rc = fm_sync_V(h)) return rc;
    int32_t *du = nullptr, *dj = nullptr, *dc = nullptr;
    double *dout = nullptr;
    hipError_t e = hipMalloc((void **)&du, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&dj, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&dc, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&dout, (size_t)n * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(du, u, (size_t)n * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dj, j, (size_t)n * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dc, ctx, (size_t)n * 4, hipMemcpyHostToDevice, h->stream);
    FmArgs a = fm_args(h);
    if (e == hipSuccess) e = fm_launch_predict(a, n, du, dj, dc, bound, lo, hi, dout, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    void *ptrs[] = {du, dj, dc, dout};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    FM_HIP(h, e);
    return CMI_OK;
}

// Recommender.evalRankings for the FM recommender (Recommender.java:668-964 with FM.predict, FM.java:93-113): same
// bookkeeping, contraction and top-N selection as cmi_eval_rankings; fp64 like the rest of the FM path.
extern "C" int cmi_fm_eval_rankings(cmi_fm_handle h, int64_t n_train, const int32_t *tu, const int32_t *tj, const int32_t *tctx,
                                    const double *tr, int64_t n_test, const int32_t *su, const int32_t *sj, const int32_t *sctx,
                                    const double *sr, double bin_thold, int num_recs, int num_ignore, int strategy,
                                    double out[CMI_RANK_MEASURES], int64_t *n_queries, int32_t *q_user, int32_t *q_ctx,
                                    int32_t *q_count, int32_t *top_items, double *top_scores) {
    if (!h) return CMI_E_INVALID;
    if (!out) FM_FAIL(h, CMI_E_INVALID, "fm_eval_rankings: null output");
    if (!h->have_model) FM_FAIL(h, CMI_E_INVALID, "fm: call cmi_fm_set_model first");
    if (n_train < 0 || n_test < 0 || (n_train > 0 && (!tu || !tj || !tctx)) || (n_test > 0 && (!su || !sj || !sctx || !sr)))
        FM_FAIL(h, CMI_E_INVALID, "fm_eval_rankings: null tuple arrays");
    if (num_recs < 1) FM_FAIL(h, CMI_E_INVALID, "fm_eval_rankings: -topN must be >= 1");
    if (strategy != CMI_RANK_UCU && strategy != CMI_RANK_UC) FM_FAIL(h, CMI_E_INVALID, "fm_eval_rankings: bad strategy");
    for (int pass = 0; pass < 2; ++pass) {
        const int64_t n = pass ? n_test : n_train;
        const int32_t *u = pass ? su : tu, *j = pass ? sj : tj, *c = pass ? sctx : tctx;
        for (int64_t t = 0; t < n; ++t)
            if (u[t] < 0 || u[t] >= h->n_users || j[t] < 0 || j[t] >= h->n_items || c[t] < 0)
                FM_FAIL(h, CMI_E_INVALID, "fm_eval_rankings: id out of range at %s tuple %lld", pass ? "test" : "train", (long long)t);
    }
    FM_HIP(h, hipSetDevice(h->device));
    if (int rc = fm_sync_V(h)) return rc;
    if (n_queries) *n_queries = 0;
    RankPlan plan;
    rank_build_plan(h->n_users, h->n_items, RankTuples{n_train, tu, tj, tctx, tr}, RankTuples{n_test, su, sj, sctx, sr}, bin_thold,
                    num_ignore, plan);
    RankWorkspace &ws = h->rank_ws;
    const int64_t nq = (int64_t)plan.qu.size();
    std::vector<double> vals((size_t)nq * 18);
    std::vector<int32_t> no_lists;
    const int32_t *top_count = nullptr;
    if (nq > 0 && !plan.cand.empty()) {
        RankOperands<double> ops;
        ops.k_logical = h->k + 1;
        const RankFmArgs base{h->d_w0, h->d_w, h->d_V, h->k, 0, h->n_users, h->n_items, h->n_conds, 1.0 / (double)h->n_ctx_dims};
        ops.build_items = [base](double *dB, const int32_t *dcand, int nc, int kp, hipStream_t s) {
            RankFmArgs a = base;
            a.kp = kp;
            return rank_launch_fm_items(a, dcand, nc, dB, s);
        };
        ops.build_queries = [base](double *dA, double *drc, const int32_t *dqu, const int32_t *dqc, int n, int kp, hipStream_t s) {
            RankFmArgs a = base;
            a.kp = kp;
            return rank_launch_fm_queries(a, dqu, dqc, n, dA, drc, s);
        };
        auto on_batch = [&](int64_t q0, int64_t q1) {
            rank_measures_range(plan, num_recs, (const int32_t *)ws.h_top.p, (const double *)ws.h_score.p, (const int32_t *)ws.h_count.p, q0, q1,
                                vals.data(), q_user, q_ctx, q_count, top_items, top_scores);
        };
        FM_HIP(h, rank_run_device<double>(h->stream, ws, plan, ops, bin_thold, num_recs, on_batch, nullptr, nullptr));
        top_count = (const int32_t *)ws.h_count.p;
    } else {
        no_lists.assign((size_t)nq, 0);
        top_count = no_lists.data();
        rank_measures_range(plan, num_recs, nullptr, nullptr, top_count, 0, nq, vals.data(), q_user, q_ctx, q_count, top_items, top_scores);
    }
    rank_average(plan, strategy, top_count, vals.data(), nullptr, RankFolded(), out);
    if (n_queries) *n_queries = (int64_t)plan.qu.size();
    return CMI_OK;
}

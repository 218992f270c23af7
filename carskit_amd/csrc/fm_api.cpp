// fm_api.cpp -- C ABI of the FM recommender (include/carskit_mi355x.h, cmi_fm_*).
#include "../../include/carskit_mi355x.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <numeric>
#include <string>
#include <vector>

#include "fm_kernels.hpp"
#include "rank_host.hpp"
#include "rank_kernels.hpp"

using namespace cmi;

struct cmi_fm_instance {
    int k = 0, n_users = 0, n_items = 0, n_conds = 0, n_ctx_dims = 1, device = 0;
    int64_t p = 0, n = 0, global_size = 0;
    std::string err;
    hipStream_t stream = nullptr;
    double *d_w0 = nullptr, *d_w = nullptr, *d_V = nullptr;
    double *d_r = nullptr, *d_part = nullptr, *d_scratch = nullptr;
    double2 *d_R = nullptr, *d_tab = nullptr;
    int32_t *d_u = nullptr, *d_j = nullptr, *d_ctx = nullptr, *d_sup[3] = {}, *d_sup_a[3] = {}, *d_sup_b[3] = {};
    bool pend_j_set = false, pend_c_set = false; // item / context deltas not yet folded into errors[]
    int col_f = -1;                               // factor whose column is loaded in d_tab[].x
    int uval_f = -1;                              // factor whose user entries are in d_R[].y
    int64_t *d_off[3] = {};
    int64_t part_count = 0;
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
    void *ptrs[] = {h->d_R, h->d_r, h->d_u, h->d_j, h->d_ctx, h->d_sup[1], h->d_sup[2], h->d_sup_a[1], h->d_sup_a[2],
                    h->d_sup_b[1], h->d_sup_b[2], h->d_off[0], h->d_off[1], h->d_off[2]};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    h->d_R = nullptr;
    h->d_r = nullptr;
    h->d_u = h->d_j = h->d_ctx = nullptr;
    for (int f = 0; f < 3; ++f) {
        h->d_sup[f] = h->d_sup_a[f] = h->d_sup_b[f] = nullptr;
        h->d_off[f] = nullptr;
    }
    h->pend_j_set = h->pend_c_set = false;
    h->have_ratings = h->initialised = false;
    h->n = 0;
}

extern "C" int cmi_fm_destroy(cmi_fm_handle h) {
    if (!h) return CMI_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    fm_free_ratings(h);
    void *ptrs[] = {h->d_w0, h->d_w, h->d_V, h->d_part, h->d_scratch, h->d_tab};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return CMI_OK;
}

extern "C" int cmi_fm_create(int k, int n_users, int n_items, int n_conds, int n_ctx_dims, int device,
                             unsigned flags, cmi_fm_handle *out) {
    (void)flags;
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
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_w, (size_t)h->p * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_V, (size_t)h->p * k * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_part, (size_t)h->part_count * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_scratch, 256 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_tab, (size_t)h->p * sizeof(double2));
    if (e == hipSuccess) e = hipMemsetAsync(h->d_tab, 0, (size_t)h->p * sizeof(double2), h->stream);
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
    h->col_f = -1;
    return CMI_OK;
}

extern "C" int cmi_fm_get_model(cmi_fm_handle h, double *w0, double *w, double *V) {
    if (!h) return CMI_E_INVALID;
    FM_HIP(h, hipSetDevice(h->device));
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

extern "C" int cmi_fm_set_ratings(cmi_fm_handle h, int64_t n, const int32_t *u, const int32_t *j, const int32_t *ctx,
                                  const double *r) {
    if (!h) return CMI_E_INVALID;
    if (n < 0 || (n > 0 && (!u || !j || !ctx || !r))) FM_FAIL(h, CMI_E_INVALID, "fm_set_ratings: null arrays");
    if (n >= ((int64_t)1 << 31)) FM_FAIL(h, CMI_E_UNSUPPORTED, "fm_set_ratings: more than 2^31-1 tuples");
    for (int64_t t = 0; t < n; ++t)
        if (u[t] < 0 || u[t] >= h->n_users || j[t] < 0 || j[t] >= h->n_items || ctx[t] < 0)
            FM_FAIL(h, CMI_E_INVALID, "fm_set_ratings: id out of range at tuple %lld", (long long)t);
    FM_HIP(h, hipSetDevice(h->device));
    FM_HIP(h, hipStreamSynchronize(h->stream));
    fm_free_ratings(h);
    // storage order: stable sort by user (counting sort) -> user supports are contiguous ranges
    std::vector<int64_t> uoff((size_t)h->n_users + 1, 0), joff((size_t)h->n_items + 1, 0), coff((size_t)h->n_conds + 1, 0);
    for (int64_t t = 0; t < n; ++t) uoff[(size_t)u[t] + 1]++;
    for (int l = 0; l < h->n_users; ++l) uoff[(size_t)l + 1] += uoff[(size_t)l];
    std::vector<int32_t> su((size_t)n), sj((size_t)n), sc((size_t)n);
    std::vector<double> sr((size_t)n);
    {
        std::vector<int64_t> cur(uoff.begin(), uoff.end() - 1);
        for (int64_t t = 0; t < n; ++t) {
            const int64_t s = cur[(size_t)u[t]]++;
            su[(size_t)s] = u[t];
            sj[(size_t)s] = j[t];
            sc[(size_t)s] = ctx[t];
            sr[(size_t)s] = r[t];
        }
    }
    // item and context-feature supports (lists of storage positions)
    for (int64_t s = 0; s < n; ++s) {
        joff[(size_t)sj[(size_t)s] + 1]++;
        if (sc[(size_t)s] < h->n_conds) coff[(size_t)sc[(size_t)s] + 1]++;
    }
    for (int l = 0; l < h->n_items; ++l) joff[(size_t)l + 1] += joff[(size_t)l];
    for (int l = 0; l < h->n_conds; ++l) coff[(size_t)l + 1] += coff[(size_t)l];
    // each entry carries the rating's other two feature ids, so a reduce pass gathers nothing but errors[]
    const size_t ncs = (size_t)coff[(size_t)h->n_conds];
    std::vector<int32_t> jsup((size_t)n), jsup_u((size_t)n), jsup_c((size_t)n), csup(ncs), csup_u(ncs), csup_j(ncs);
    {
        std::vector<int64_t> cj(joff.begin(), joff.end() - 1), cc(coff.begin(), coff.end() - 1);
        for (int64_t s = 0; s < n; ++s) {
            const size_t pj = (size_t)cj[(size_t)sj[(size_t)s]]++;
            jsup[pj] = (int32_t)s;
            jsup_u[pj] = su[(size_t)s];
            jsup_c[pj] = sc[(size_t)s];
            if (sc[(size_t)s] < h->n_conds) {
                const size_t pc = (size_t)cc[(size_t)sc[(size_t)s]]++;
                csup[pc] = (int32_t)s;
                csup_u[pc] = su[(size_t)s];
                csup_j[pc] = sj[(size_t)s];
            }
        }
    }
    hipError_t e = up(&h->d_u, su, h->stream);
    if (e == hipSuccess) e = up(&h->d_j, sj, h->stream);
    if (e == hipSuccess) e = up(&h->d_ctx, sc, h->stream);
    if (e == hipSuccess) e = up(&h->d_r, sr, h->stream);
    if (e == hipSuccess) e = up(&h->d_sup[1], jsup, h->stream);
    if (e == hipSuccess) e = up(&h->d_sup[2], csup, h->stream);
    if (e == hipSuccess) e = up(&h->d_sup_a[1], jsup_u, h->stream);
    if (e == hipSuccess) e = up(&h->d_sup_b[1], jsup_c, h->stream);
    if (e == hipSuccess) e = up(&h->d_sup_a[2], csup_u, h->stream);
    if (e == hipSuccess) e = up(&h->d_sup_b[2], csup_j, h->stream);
    if (e == hipSuccess) e = up(&h->d_off[0], uoff, h->stream);
    if (e == hipSuccess) e = up(&h->d_off[1], joff, h->stream);
    if (e == hipSuccess) e = up(&h->d_off[2], coff, h->stream);
    if (e == hipSuccess && n > 0) e = hipMalloc((void **)&h->d_R, (size_t)n * sizeof(double2));
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        fm_free_ratings(h);
        FM_FAIL(h, CMI_E_HIP, "fm_set_ratings: upload failed: %s", hipGetErrorString(e));
    }
    h->n = n;
    h->have_ratings = true;
    return CMI_OK;
}

static FmArgs fm_args(cmi_fm_instance *h) {
    FmArgs a;
    a.w0 = h->d_w0;
    a.w = h->d_w;
    a.V = h->d_V;
    a.R = h->d_R;
    a.tab = h->d_tab;
    a.pending = (h->pend_j_set ? 1 : 0) | (h->pend_c_set ? 2 : 0);
    a.u = h->d_u;
    a.j = h->d_j;
    a.ctx = h->d_ctx;
    a.r = h->d_r;
    for (int f = 0; f < 3; ++f) {
        a.sup[f] = h->d_sup[f];
        a.sup_a[f] = h->d_sup_a[f];
        a.sup_b[f] = h->d_sup_b[f];
    }
    for (int f = 0; f < 3; ++f) a.sup_off[f] = h->d_off[f];
    a.field_count[0] = h->n_users;
    a.field_count[1] = h->n_items;
    a.field_count[2] = h->n_conds;
    a.part = h->d_part;
    a.n = h->n;
    a.global_size = h->global_size > 0 ? h->global_size : h->n;
    a.k = h->k;
    a.n_users = h->n_users;
    a.n_items = h->n_items;
    a.n_conds = h->n_conds;
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
    h->pend_j_set = h->pend_c_set = false;
    h->col_f = h->uval_f = -1;
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

// ---- phase driver: keeps the lazy-error bookkeeping consistent for any call order ---------------------------------
// Usual order (w0, then per factor: users, items, contexts): the user phase and the w0 phase fold the pending item /
// context deltas into errors[] on their own sequential pass, so no extra pass is ever launched.
static int fm_before_phase(cmi_fm_instance *h, int field, int f) {
    if (f >= 0 && h->col_f != f) {
        FM_HIP(h, fm_launch_col_load(fm_args(h), f, h->stream));
        h->col_f = f;
    }
    if (f >= 0 && field != 0 && h->uval_f != f) { // only when driven out of order: the user phase of f leaves them there
        FM_HIP(h, fm_launch_uval(fm_args(h), h->stream));
        h->uval_f = f;
    }
    // an item phase overwrites pend_j and a context phase pend_c: fold first if they still hold deltas
    if ((field == 1 && (h->pend_j_set || h->pend_c_set)) || (field == 2 && h->pend_c_set)) {
        FM_HIP(h, fm_launch_flush(fm_args(h), h->stream));
        h->pend_j_set = h->pend_c_set = false;
    }
    return CMI_OK;
}

static int fm_after_apply(cmi_fm_instance *h, int field, int f = -1) {
    if (field == 0 && f >= 0) h->uval_f = f;
    if (field <= 0) h->pend_j_set = h->pend_c_set = false; // w0 (-1) and user (0) passes rewrote errors[] with the deltas folded in
    else if (field == 1) h->pend_j_set = true;
    else h->pend_c_set = true;
    return CMI_OK;
}

extern "C" int cmi_fm_phase_reduce(cmi_fm_handle h, int phase) {
    if (!h) return CMI_E_INVALID;
    if (int rc = fm_ready(h, true)) return rc;
    int field, f;
    if (!phase_decode(h, phase, &field, &f)) FM_FAIL(h, CMI_E_INVALID, "fm: bad phase %d", phase);
    if (int rc = fm_before_phase(h, field, f)) return rc;
    const FmArgs a = fm_args(h);
    if (phase == 0) FM_HIP(h, fm_launch_w0_reduce(a, h->d_scratch, h->stream));
    else FM_HIP(h, fm_launch_field(a, field, f, 0, h->stream));
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
    const FmArgs a = fm_args(h);
    if (phase == 0) FM_HIP(h, fm_launch_w0_apply(a, h->stream));
    else FM_HIP(h, fm_launch_field(a, field, f, 1, h->stream));
    return fm_after_apply(h, field, f);
}

// one phase, reduce + update fused (no exchange point): what cmi_fm_sweep runs per phase; a multi-GPU host uses it for the
// phases whose coordinates live on one rank only (the user field of user-sharded ratings)
extern "C" int cmi_fm_phase_run(cmi_fm_handle h, int phase) {
    if (!h) return CMI_E_INVALID;
    if (int rc = fm_ready(h, true)) return rc;
    int field, f;
    if (!phase_decode(h, phase, &field, &f)) FM_FAIL(h, CMI_E_INVALID, "fm: bad phase %d", phase);
    if (int rc = fm_before_phase(h, field, f)) return rc;
    if (phase == 0) {
        FM_HIP(h, fm_launch_w0_reduce(fm_args(h), h->d_scratch, h->stream));
        FM_HIP(h, fm_launch_w0_apply(fm_args(h), h->stream));
    } else {
        FM_HIP(h, fm_launch_field(fm_args(h), field, f, 2, h->stream));
    }
    h->last_phase = -1;
    return fm_after_apply(h, field, f);
}

extern "C" int cmi_fm_sweep(cmi_fm_handle h) {
    if (!h) return CMI_E_INVALID;
    if (int rc = fm_ready(h, true)) return rc;
    FM_HIP(h, fm_launch_w0_reduce(fm_args(h), h->d_scratch, h->stream));
    FM_HIP(h, fm_launch_w0_apply(fm_args(h), h->stream));
    fm_after_apply(h, -1);
    for (int f = -1; f < h->k; ++f)
        for (int field = 0; field < 3; ++field) {
            if (int rc = fm_before_phase(h, field, f)) return rc;
            FM_HIP(h, fm_launch_field(fm_args(h), field, f, 2, h->stream));
            fm_after_apply(h, field, f);
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
    if (n_queries) *n_queries = 0;
    RankPlan plan;
    rank_build_plan(h->n_users, h->n_items, RankTuples{n_train, tu, tj, tctx, tr}, RankTuples{n_test, su, sj, sctx, sr}, bin_thold,
                    num_ignore, plan);
    std::vector<int32_t> top_idx, top_count;
    std::vector<double> top_score;
    if (!plan.qu.empty() && !plan.cand.empty()) {
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
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        hipError_t e = hipEventCreate(&ev0);
        if (e == hipSuccess) e = hipEventCreate(&ev1);
        if (e == hipSuccess)
            e = rank_run_device<double>(h->stream, ev0, ev1, plan, ops, bin_thold, num_recs, top_idx, top_score, top_count, nullptr, nullptr);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        FM_HIP(h, e);
    } else {
        top_count.assign(plan.qu.size(), 0);
    }
    rank_metrics(plan, strategy, num_recs, top_idx, top_score, top_count, out, q_user, q_ctx, q_count, top_items, top_scores);
    if (n_queries) *n_queries = (int64_t)plan.qu.size();
    return CMI_OK;
}

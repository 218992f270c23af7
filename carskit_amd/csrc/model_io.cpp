// model_io.cpp -- cmi_save_model / cmi_load_model: persistence of a trained recommender.
//
// The reference's IterativeRecommender.saveModel()/loadModel() (src/carskit/generic/IterativeRecommender.java:249-292) write
// P, Q, userBias, itemBias as Java object-serialization streams of librec classes, one file each, and FORGET the context
// containers (condBias, ucBias, icBias), so a reloaded CAMF model predicts without its context deviations.  Without a JVM such a
// stream cannot be produced verifiably; this library keeps its own documented, versioned, self-checking format instead and
// stores everything predict() needs.
//
// File layout (little endian):
//   0   char[8]  magic "CMIMODL1"
//   8   u32      format version (2; version-1 files, which lack the block at 96, are still read)
//   12  u32      model id (CMI_MODEL_*)
//   16  u32 k, u32 n_users, u32 n_items, u32 n_conds
//   32  f64      globalMean, regU, regI, regB, regC            (what predict()/a resumed buildModel() need besides the tables)
//   72  f64      lRate, last_loss ; u32 epochs_done, u32 n_containers      (resume state of the epoch loop)
//   96  (version 2) u32 numF, u32 n_ctx_dims, u32 n_empty, i32 empty_conds[n_empty]: what predict() of CAMF_ICS / LCS / MCS needs
//       besides the tables (cmi_set_sim_params).  On load they are RESTORED into a handle that has none yet and VERIFIED against a
//       handle that has (a model trained with other EmptyContextConditions must not pass unnoticed: ADVICE r2)
//   ..  per container: u32 which (CMI_STATE_*), u32 reserved, u64 count, then count f64 values (row-major, as cmi_get_state)
//   end u64      FNV-1a 64 of every preceding byte
// cmi_load_model also OVERWRITES the handle's regularisers and globalMean with the stored ones (a resumed buildModel() must use the
// hyper-parameters the model was trained with).
// Values are always stored as fp64: lossless for both state dtypes (fp32 state widens exactly and narrows back exactly).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "cmi_instance.hpp"

namespace {

const char kMagic[8] = {'C', 'M', 'I', 'M', 'O', 'D', 'L', '1'};

struct Fnv {
    uint64_t h = 1469598103934665603ull;
    void add(const void *p, size_t n) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < n; ++i) {
            h ^= b[i];
            h *= 1099511628211ull;
        }
    }
};

struct Writer {
    FILE *f;
    Fnv sum;
    bool ok = true;
    void put(const void *p, size_t n) {
        if (ok && fwrite(p, 1, n, f) != n) ok = false;
        sum.add(p, n);
    }
    void u32(uint32_t v) { put(&v, 4); }
    void u64(uint64_t v) { put(&v, 8); }
    void f64(double v) { put(&v, 8); }
};

struct Reader {
    FILE *f;
    Fnv sum;
    bool ok = true;
    void get(void *p, size_t n) {
        if (ok && fread(p, 1, n, f) != n) ok = false;
        if (ok) sum.add(p, n);
    }
    uint32_t u32() {
        uint32_t v = 0;
        get(&v, 4);
        return v;
    }
    uint64_t u64() {
        uint64_t v = 0;
        get(&v, 8);
        return v;
    }
    double f64() {
        double v = 0;
        get(&v, 8);
        return v;
    }
};

} // namespace

extern "C" int cmi_save_model(cmi_handle h, const char *path, double lrate, double last_loss, int epochs_done) {
    if (!h || !path) return CMI_E_INVALID;
    FILE *f = fopen(path, "wb");
    if (!f) CMI_FAIL(h, CMI_E_INVALID, "save_model: cannot open %s for writing", path);
    Writer w{f};
    w.put(kMagic, 8);
    w.u32(2);
    w.u32((uint32_t)h->model);
    w.u32((uint32_t)h->k);
    w.u32((uint32_t)h->n_users);
    w.u32((uint32_t)h->n_items);
    w.u32((uint32_t)h->n_conds);
    w.f64(h->hp.gm);
    w.f64(h->hp.regU);
    w.f64(h->hp.regI);
    w.f64(h->hp.regB);
    w.f64(h->hp.regC);
    w.f64(lrate);
    w.f64(last_loss);
    w.u32((uint32_t)epochs_done);
    uint32_t nc = 0;
    for (int c = 0; c < CMI_STATE_COUNT; ++c)
        if (cmi_model_has(h->model, c)) ++nc;
    w.u32(nc);
    w.u32((uint32_t)h->num_f);
    w.u32((uint32_t)h->n_ctx_dims);
    w.u32((uint32_t)h->empty_conds.size());
    if (!h->empty_conds.empty()) w.put(h->empty_conds.data(), h->empty_conds.size() * 4);
    std::vector<double> buf;
    int rc = CMI_OK;
    for (int c = 0; c < CMI_STATE_COUNT && rc == CMI_OK; ++c) {
        if (!cmi_model_has(h->model, c)) continue;
        buf.resize((size_t)h->state_count[c]);
        if (!buf.empty()) rc = cmi_get_state(h, c, buf.data(), h->state_count[c], CMI_DTYPE_F64);
        w.u32((uint32_t)c);
        w.u32(0);
        w.u64((uint64_t)h->state_count[c]);
        w.put(buf.data(), buf.size() * 8);
    }
    const uint64_t sum = w.sum.h;
    if (w.ok && fwrite(&sum, 1, 8, f) != 8) w.ok = false;
    if (fclose(f) != 0) w.ok = false;
    if (rc != CMI_OK) return rc;
    if (!w.ok) CMI_FAIL(h, CMI_E_INVALID, "save_model: short write to %s", path);
    return CMI_OK;
}

extern "C" int cmi_load_model(cmi_handle h, const char *path, double *lrate, double *last_loss, int *epochs_done) {
    if (!h || !path) return CMI_E_INVALID;
    FILE *f = fopen(path, "rb");
    if (!f) CMI_FAIL(h, CMI_E_INVALID, "load_model: cannot open %s", path);
    Reader r{f};
    char magic[8];
    r.get(magic, 8);
    const uint32_t version = r.u32(), model = r.u32(), k = r.u32(), nu = r.u32(), ni = r.u32(), ncd = r.u32();
#define LOAD_FAIL(...)            \
    do {                          \
        fclose(f);                \
        CMI_FAIL(h, CMI_E_INVALID, __VA_ARGS__); \
    } while (0)
    if (!r.ok || memcmp(magic, kMagic, 8) != 0) LOAD_FAIL("load_model: %s is not a CMIMODL1 file", path);
    if (version != 1 && version != 2) LOAD_FAIL("load_model: format version %u not supported (this library reads versions 1 and 2)", version);
    if ((int)model != h->model || (int)k != h->k || (int)nu != h->n_users || (int)ni != h->n_items || (int)ncd != h->n_conds)
        LOAD_FAIL("load_model: file holds model %u k=%u %ux%u users/items %u conditions, the handle is model %d k=%d %dx%d %d", model, k,
                  nu, ni, ncd, h->model, h->k, h->n_users, h->n_items, h->n_conds);
    double hp[5];
    for (double &v : hp) v = r.f64();
    const double lr = r.f64(), ll = r.f64();
    const uint32_t ep = r.u32(), nc = r.u32();
    uint32_t f_numf = 0, f_dims = 1;
    std::vector<int32_t> f_empty;
    if (version >= 2) {
        f_numf = r.u32();
        f_dims = r.u32();
        const uint32_t ne = r.u32();
        if (!r.ok || ne > (uint32_t)h->n_conds + 1024u) LOAD_FAIL("load_model: %s is truncated or corrupt (sim-parameter block)", path);
        f_empty.resize(ne);
        if (ne) r.get(f_empty.data(), (size_t)ne * 4);
        if (h->sim_params_set && (f_empty != h->empty_conds || (int)f_numf != h->num_f || (int)f_dims != h->n_ctx_dims))
            LOAD_FAIL("load_model: %s was trained with other EmptyContextConditions / numF / context dimensions than this handle has "
                      "(cmi_set_sim_params)", path);
    }
    // a version-2 file carries the sim parameters: a handle that has none yet takes them from the file -- also when the list of
    // empty conditions is EMPTY (the restore used to be skipped then, and CAMF_LCS's cfMatrix failed its count check
    // after P and Q had already been overwritten)
    const bool sim_model = h->model == CMI_MODEL_CAMF_ICS || h->model == CMI_MODEL_CAMF_LCS || h->model == CMI_MODEL_CAMF_MCS;
    const bool restore_sim = version >= 2 && sim_model && !h->sim_params_set;
    if (restore_sim && (f_dims < 1 || (h->model == CMI_MODEL_CAMF_LCS && f_numf < 1)))
        LOAD_FAIL("load_model: %s holds invalid sim parameters (numF %u, %u context dimensions)", path, f_numf, f_dims);
    for (int32_t c : f_empty)
        if (c < 0 || c >= h->n_conds) LOAD_FAIL("load_model: %s lists an empty condition id out of range", path);
    std::vector<std::vector<double>> tabs(CMI_STATE_COUNT);
    std::vector<bool> seen(CMI_STATE_COUNT, false);
    for (uint32_t i = 0; i < nc && r.ok; ++i) {
        const uint32_t which = r.u32();
        (void)r.u32();
        const uint64_t count = r.u64();
        // a handle without sim params yet gets them from the file (below): its cfMatrix then has n_conds x numF elements
        int64_t expect = which < CMI_STATE_COUNT ? h->state_count[which] : -1;
        if (which == CMI_STATE_CF_MATRIX && restore_sim) expect = (int64_t)h->n_conds * (int64_t)f_numf;
        if (!r.ok || which >= CMI_STATE_COUNT || !cmi_model_has(h->model, (int)which) || (int64_t)count != expect || seen[which])
            LOAD_FAIL("load_model: unexpected container %u (count %llu) in %s", which, (unsigned long long)count, path);
        seen[which] = true;
        tabs[which].resize((size_t)count);
        r.get(tabs[which].data(), (size_t)count * 8);
    }
    const uint64_t want = r.sum.h;
    uint64_t got = 0;
    const bool have_sum = r.ok && fread(&got, 1, 8, f) == 8;
    if (!have_sum || got != want) LOAD_FAIL("load_model: %s is truncated or corrupt (checksum mismatch)", path);
    for (int c = 0; c < CMI_STATE_COUNT; ++c)
        if (cmi_model_has(h->model, c) && !seen[c]) LOAD_FAIL("load_model: container %d missing in %s", c, path);
#undef LOAD_FAIL
    fclose(f);
    // only now touch the handle: a bad file leaves the model as it was.  Every count was validated above against the sizes the
    // handle has AFTER the sim-parameter restore, and the restore's own arguments were checked, so what can still fail below is a
    // HIP call, not the file.
    if (restore_sim)
        if (int rc = cmi_set_sim_params(h, (int)f_numf, (int)f_dims, f_empty.empty() ? nullptr : f_empty.data(), (int)f_empty.size())) return rc;
    for (int c = 0; c < CMI_STATE_COUNT; ++c)
        if (seen[c] && !tabs[c].empty())
            if (int rc = cmi_set_state(h, c, tabs[c].data(), (int64_t)tabs[c].size(), CMI_DTYPE_F64)) return rc;
    cmi_set_hparams(h, hp[1], hp[2], hp[3], hp[4], hp[0]);
    if (lrate) *lrate = lr;
    if (last_loss) *last_loss = ll;
    if (epochs_done) *epochs_done = (int)ep;
    return CMI_OK;
}

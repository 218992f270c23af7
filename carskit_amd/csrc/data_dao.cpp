// data_dao.cpp -- host-side integer path either side of the hot loop:
//   * cmi_dao_read: the id-mapper of DataDAO.readData (reference src/carskit/data/processor/DataDAO.java:166-354)
//     for the *binary* rating format: first-seen inner ids for users / items / (user,item) pairs / context
//     combinations, the condition table, and the (user-item x context) rating matrix in the CRS order librec's
//     MatrixIterator yields.  Must be bit-exact (north_star: "integer id mapping bit-exact").
//   * cmi_transform_compact_to_binary: DataTransformer.TransformationFromCompactToBinary
//     (src/carskit/data/processor/DataTransformer.java:231-259, 266-329) including the row order of the
//     rewritten train.csv, which is the iteration order of a java.util.HashMap<String,...> keyed by the
//     input line (String.hashCode, hash spreading h ^ (h>>>16), power-of-two table grown at load 0.75,
//     buckets in index order, insertion order inside a bucket).
// Plain C++ (no GPU involved); exported through the same C ABI so the Java/ctypes hosts and the tests share it.
#include "../../include/carskit_mi355x.h"

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

// java.lang.String.trim(): strip code units <= U+0020 from both ends
std::string jtrim(const std::string &s) {
    size_t b = 0, e = s.size();
    while (b < e && (unsigned char)s[b] <= 0x20) ++b;
    while (e > b && (unsigned char)s[e - 1] <= 0x20) --e;
    return s.substr(b, e - b);
}

std::string jlower(std::string s) { // String.toLowerCase() for the ASCII range (data files are ASCII)
    for (char &c : s)
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    return s;
}

// String.split(",", -1): every field kept, trailing empties included
std::vector<std::string> split_keep(const std::string &s, char d) {
    std::vector<std::string> out;
    size_t b = 0;
    while (true) {
        size_t p = s.find(d, b);
        if (p == std::string::npos) {
            out.push_back(s.substr(b));
            break;
        }
        out.push_back(s.substr(b, p - b));
        b = p + 1;
    }
    return out;
}

// String.split("[\t,]+") (limit 0): runs of tab/comma separate; a leading empty string is kept when the input
// starts with a separator; trailing empty strings are removed
std::vector<std::string> split_runs(const std::string &s) {
    std::vector<std::string> out;
    size_t i = 0, n = s.size();
    std::string cur;
    bool any_sep = false;
    while (i < n) {
        if (s[i] == '\t' || s[i] == ',') {
            size_t j = i;
            while (j < n && (s[j] == '\t' || s[j] == ',')) ++j;
            out.push_back(cur);
            cur.clear();
            any_sep = true;
            i = j;
        } else {
            cur.push_back(s[i++]);
        }
    }
    out.push_back(cur);
    if (!any_sep) return out; // no match: the whole string
    while (!out.empty() && out.back().empty()) out.pop_back();
    return out;
}

// BufferedReader.readLine(): \n, \r or \r\n terminate a line; a final unterminated line counts
bool read_lines(const char *path, std::vector<std::string> &lines, std::string &err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        err = std::string("cannot open ") + path;
        return false;
    }
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string all = ss.str();
    size_t i = 0, n = all.size();
    std::string cur;
    bool pending = false;
    while (i < n) {
        const char c = all[i++];
        if (c == '\n' || c == '\r') {
            if (c == '\r' && i < n && all[i] == '\n') ++i;
            lines.push_back(cur);
            cur.clear();
            pending = false;
        } else {
            cur.push_back(c);
            pending = true;
        }
    }
    if (pending) lines.push_back(cur);
    return true;
}

// the same line structure without one std::string per line: (offset, length) spans into the file image
bool read_spans(const char *path, std::string &all, std::vector<std::pair<size_t, size_t>> &spans, std::string &err) {
    FILE *f = std::fopen(path, "rb");
    if (!f) {
        err = std::string("cannot open ") + path;
        return false;
    }
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    all.resize(sz > 0 ? (size_t)sz : 0);
    const size_t got = all.empty() ? 0 : std::fread(&all[0], 1, all.size(), f);
    std::fclose(f);
    all.resize(got);
    size_t i = 0, b = 0;
    const size_t n = all.size();
    while (i < n) {
        const char c = all[i];
        if (c == '\n' || c == '\r') {
            spans.emplace_back(b, i - b);
            ++i;
            if (c == '\r' && i < n && all[i] == '\n') ++i;
            b = i;
        } else {
            ++i;
        }
    }
    if (b < n) spans.emplace_back(b, n - b);
    return true;
}

// Double.valueOf(String): optional surrounding whitespace, optional trailing f/F/d/D, decimal or hex float,
// "NaN", "Infinity" with optional sign
bool jparse_double(const std::string &raw, double &out) {
    std::string s = jtrim(raw);
    if (s.empty()) return false;
    if (s.size() > 1) {
        const char last = s.back();
        if (last == 'f' || last == 'F' || last == 'd' || last == 'D') {
            const bool hex = s.find("0x") != std::string::npos || s.find("0X") != std::string::npos;
            if (!hex || s.find_first_of("pP") != std::string::npos) s.pop_back();
        }
    }
    std::string body = s;
    int sign = 1;
    if (body[0] == '+' || body[0] == '-') {
        sign = body[0] == '-' ? -1 : 1;
        body = body.substr(1);
    }
    if (body == "NaN") {
        out = std::nan("");
        return true;
    }
    if (body == "Infinity") {
        out = sign * HUGE_VAL;
        return true;
    }
    if (body.empty() || !(std::isdigit((unsigned char)body[0]) || body[0] == '.')) return false;
    for (char c : body) // Java accepts no "inf"/"nan"/"infinity" spellings of strtod and no embedded spaces
        if (std::isspace((unsigned char)c)) return false;
    errno = 0;
    char *end = nullptr;
    out = std::strtod(s.c_str(), &end);
    return end && *end == '\0' && end != s.c_str();
}

// Integer.valueOf(String): optional sign then decimal digits, nothing else, within int32
bool jparse_int(const std::string &s, int32_t &out) {
    if (s.empty()) return false;
    size_t i = 0;
    bool neg = false;
    if (s[0] == '+' || s[0] == '-') {
        neg = s[0] == '-';
        i = 1;
    }
    if (i >= s.size()) return false;
    int64_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (s[i] - '0');
        if (v > (int64_t)1 << 31) return false;
    }
    v = neg ? -v : v;
    if (v > INT32_MAX || v < INT32_MIN) return false;
    out = (int32_t)v;
    return true;
}

// first-seen id assignment keyed by raw byte strings (HashMap<String,Integer>.containsKey / put(key, size()) of the
// reference, DataDAO.java:237-241,266-268,332-333): open addressing over the names vector, no per-lookup allocation
class StrIndex {
  public:
    int32_t find_or_add(const char *p, size_t n, std::vector<std::string> &names) {
        if ((names.size() + 1) * 2 > slots_.size()) grow(names);
        const uint64_t h = hash(p, n);
        size_t i = (size_t)h & (slots_.size() - 1);
        while (slots_[i].id >= 0) {
            if (slots_[i].h == h) {
                const std::string &s = names[(size_t)slots_[i].id];
                if (s.size() == n && std::memcmp(s.data(), p, n) == 0) return slots_[i].id;
            }
            i = (i + 1) & (slots_.size() - 1);
        }
        const int32_t id = (int32_t)names.size();
        names.emplace_back(p, n);
        slots_[i] = Slot{h, id};
        return id;
    }
    void rebuild(const std::vector<std::string> &names) { // after the names were copied from another DAO
        slots_.clear();
        grow(names);
    }

  private:
    static uint64_t hash(const char *p, size_t n) { // FNV-1a, 64 bit
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < n; ++i) h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
        return h ^ (h >> 29);
    }
    void grow(const std::vector<std::string> &names) {
        size_t cap = slots_.empty() ? 1024 : slots_.size() * 2;
        while (cap < (names.size() + 1) * 2) cap *= 2;
        slots_.assign(cap, Slot{0, -1});
        for (size_t id = 0; id < names.size(); ++id) {
            const uint64_t h = hash(names[id].data(), names[id].size());
            size_t i = (size_t)h & (cap - 1);
            while (slots_[i].id >= 0) i = (i + 1) & (cap - 1);
            slots_[i] = Slot{h, (int32_t)id};
        }
    }
    struct Slot {
        uint64_t h;
        int32_t id;
    };
    std::vector<Slot> slots_;
};

// (user, item) pair -> ui id, first seen (the reference keys a HashMap by the string "<u>,<i>" of inner ids)
class PairIndex {
  public:
    // returns the id; *fresh tells whether the pair was new (then id == previous size)
    int32_t find_or_add(uint64_t key, int32_t next_id, bool *fresh) {
        if ((size_ + 1) * 2 > slots_.size()) grow();
        size_t i = (size_t)mix(key) & (slots_.size() - 1);
        while (slots_[i].id >= 0) {
            if (slots_[i].key == key) {
                *fresh = false;
                return slots_[i].id;
            }
            i = (i + 1) & (slots_.size() - 1);
        }
        slots_[i] = Slot{key, next_id};
        ++size_;
        *fresh = true;
        return next_id;
    }

  private:
    static uint64_t mix(uint64_t x) {
        x ^= x >> 33;
        x *= 0xff51afd7ed558ccdull;
        x ^= x >> 33;
        return x;
    }
    struct Slot {
        uint64_t key;
        int32_t id;
    };
    void grow() {
        std::vector<Slot> old;
        old.swap(slots_);
        slots_.assign(old.empty() ? 1024 : old.size() * 2, Slot{0, -1});
        for (const Slot &s : old)
            if (s.id >= 0) {
                size_t i = (size_t)mix(s.key) & (slots_.size() - 1);
                while (slots_[i].id >= 0) i = (i + 1) & (slots_.size() - 1);
                slots_[i] = s;
            }
    }
    std::vector<Slot> slots_;
    size_t size_ = 0;
};

template <typename M>
int32_t first_seen(M &m, std::vector<std::string> &names, const std::string &key) {
    auto it = m.find(key);
    if (it != m.end()) return it->second;
    const int32_t id = (int32_t)names.size();
    m.emplace(key, id);
    names.push_back(key);
    return id;
}

} // namespace

struct cmi_dao {
    std::string err;
    std::unordered_map<std::string, int32_t> dim_ids;
    StrIndex user_ids, item_ids, ctx_ids;           // first-seen ids over users / items / ctxs
    PairIndex ui_ids;                               // (user << 32 | item) -> ui id; the reference keys the string "u,i"
    std::vector<std::string> users, items, ctxs, dims, conds; // inner id -> raw key
    mutable std::string scratch;                    // cmi_dao_raw_id(kind 5) formats "u,i" on demand
    std::vector<int32_t> ui_user, ui_item, cond_dim, empty_conds;
    std::vector<std::vector<int32_t>> ctx_cond_list;
    std::vector<double> rating_scale;
    int64_t num_ratings = 0; // lines (scaleDist.size())
    // matrix, CRS order
    std::vector<int32_t> m_ui, m_ctx;
    std::vector<double> m_r;
};

static thread_local std::string g_dao_err;

extern "C" const char *cmi_dao_last_error(cmi_dao_handle h) { return h ? h->err.c_str() : g_dao_err.c_str(); }

extern "C" int cmi_dao_destroy(cmi_dao_handle h) {
    delete h;
    return CMI_OK;
}

static int dao_read_impl(const char *path, const cmi_dao *base, cmi_dao_handle *out) {
    if (out) *out = nullptr;
    if (!path || !out) {
        g_dao_err = "cmi_dao_read: null argument";
        return CMI_E_INVALID;
    }
    std::string image;
    std::vector<std::pair<size_t, size_t>> lines; // spans into `image`
    if (!read_spans(path, image, lines, g_dao_err)) return CMI_E_INVALID;
    if (lines.empty()) {
        g_dao_err = "cmi_dao_read: empty file (the reference dereferences a null header line)";
        return CMI_E_INVALID;
    }
    cmi_dao *d = new cmi_dao();
    if (base) { // `test-set`: the test DAO is constructed over the TRAIN DAO's maps (CARSKit.java:335-340) and extends them
        d->ui_ids = base->ui_ids;
        d->dim_ids = base->dim_ids;
        d->users = base->users;
        d->items = base->items;
        d->ctxs = base->ctxs;
        d->user_ids.rebuild(d->users);
        d->item_ids.rebuild(d->items);
        d->ctx_ids.rebuild(d->ctxs);
        d->dims = base->dims;
        d->conds = base->conds;
        d->cond_dim = base->cond_dim;
        d->ui_user = base->ui_user;
        d->ui_item = base->ui_item;
        d->ctx_cond_list = base->ctx_cond_list;
    }
    // header (DataDAO.java:198-215): trim, split on runs of tab/comma, columns >= 3 are conditions
    {
        const std::vector<std::string> hd = split_runs(jtrim(image.substr(lines[0].first, lines[0].second)));
        for (size_t i = 3; i < hd.size(); ++i) {
            const std::string context = jtrim(hd[i]);
            const size_t colon = context.find(':');
            // context.split(":")[0]: text before the first ':' (an empty first token stays empty)
            const std::string dim = jtrim(colon == std::string::npos ? context : context.substr(0, colon));
            const int32_t dimc = first_seen(d->dim_ids, d->dims, dim);
            if (i - 3 < d->conds.size()) {
                // condIds.put(context, i-3) on the shared BiMap: a different token for an existing column would
                // throw IllegalArgumentException (value already present)
                if (d->conds[i - 3] != context) {
                    g_dao_err = "test header column " + std::to_string(i) + " ('" + context + "') differs from the training header ('" + d->conds[i - 3] + "')";
                    delete d;
                    return CMI_E_INVALID;
                }
            } else {
                d->conds.push_back(context);
                d->cond_dim.push_back(dimc);
            }
            const std::string na = ":na";
            if (context.size() >= na.size() && context.compare(context.size() - na.size(), na.size(), na) == 0)
                d->empty_conds.push_back((int32_t)i - 3);
        }
    }
    const int32_t n_conds = (int32_t)d->conds.size();
    // data lines (DataDAO.java:222-345); dataTable.put(uic, cc, rate): the LAST line of a (ui, ctx) cell wins.
    // Fields are scanned in place (no per-line allocations); the general Double.valueOf / Integer.valueOf restatements
    // are only called for tokens that are not plain digits.
    struct Cell {
        uint64_t key; // ui << 32 | ctx  (CRS order = ascending key)
        double rate;
    };
    std::vector<Cell> cells;
    cells.reserve(lines.size());
    std::vector<double> scale;
    std::vector<int32_t> cond_list;
    std::string ctx;
    for (size_t ln = 1; ln < lines.size(); ++ln) {
        const char *const raw = image.data() + lines[ln].first;
        // line.trim(): strip chars <= ' ' at both ends
        size_t lb = 0, le = lines[ln].second;
        while (lb < le && (unsigned char)raw[lb] <= ' ') ++lb;
        while (le > lb && (unsigned char)raw[le - 1] <= ' ') --le;
        const char *p = raw + lb, *const end = raw + le;
        // split(",", -1): every comma separates, empties kept
        auto next_field = [&](const char *&fb, const char *&fe) -> bool {
            if (p > end) return false;
            fb = p;
            const char *c = (const char *)std::memchr(p, ',', (size_t)(end - p));
            fe = c ? c : end;
            p = fe + 1; // one past the comma; > end after the last field
            return true;
        };
        const char *ub, *ue, *ib, *ie, *rb, *re;
        if (!next_field(ub, ue) || !next_field(ib, ie) || !next_field(rb, re)) {
            d->err = "line " + std::to_string(ln + 1) + ": fewer than 3 fields (ArrayIndexOutOfBounds in the reference)";
            g_dao_err = d->err;
            delete d;
            return CMI_E_INVALID;
        }
        double rate = 0.0;
        {
            bool simple = re > rb && re - rb <= 15; // [0-9]+ ( . [0-9]+ )? : exact in double for <= 15 digits
            int64_t ip = 0, fp = 0, fdig = 0;
            const char *c = rb;
            for (; simple && c < re && *c >= '0' && *c <= '9'; ++c) ip = ip * 10 + (*c - '0');
            if (simple && c == rb) simple = false;
            if (simple && c < re && *c == '.') {
                ++c;
                const char *f0 = c;
                for (; c < re && *c >= '0' && *c <= '9'; ++c, ++fdig) fp = fp * 10 + (*c - '0');
                if (c == f0) simple = false;
            }
            if (simple && c == re && fdig == 0) rate = (double)ip;
            else if (!jparse_double(std::string(rb, re), rate)) { // everything else: the Double.valueOf restatement
                g_dao_err = "line " + std::to_string(ln + 1) + ": rating '" + std::string(rb, re) + "' is not a number (NumberFormatException)";
                delete d;
                return CMI_E_INVALID;
            }
        }
        scale.push_back(rate);
        d->num_ratings++;
        const int32_t row = d->user_ids.find_or_add(ub, (size_t)(ue - ub), d->users); // NOT trimmed (DataDAO.java:226-227)
        const int32_t col = d->item_ids.find_or_add(ib, (size_t)(ie - ib), d->items);
        const uint64_t uikey = ((uint64_t)(uint32_t)row << 32) | (uint32_t)col;
        bool fresh = false;
        const int32_t uic = d->ui_ids.find_or_add(uikey, (int32_t)d->ui_user.size(), &fresh);
        if (fresh) {
            d->ui_user.push_back(row);
            d->ui_item.push_back(col);
        }
        ctx.clear();
        cond_list.clear();
        const char *fb, *fe;
        for (int32_t ci = 0; next_field(fb, fe); ++ci) {
            int32_t value;
            while (fb < fe && (unsigned char)*fb <= ' ') ++fb; // data[i].trim()
            while (fe > fb && (unsigned char)fe[-1] <= ' ') --fe;
            if (fe - fb == 1 && (*fb == '0' || *fb == '1')) value = *fb - '0';
            else if (!jparse_int(std::string(fb, fe), value)) {
                g_dao_err = "line " + std::to_string(ln + 1) + ": condition flag '" + std::string(fb, fe) + "' is not an integer (NumberFormatException)";
                delete d;
                return CMI_E_INVALID;
            }
            if (value == 1) {
                if (!ctx.empty()) ctx += ',';
                char num[12];
                int nd = 0, v = ci;
                do num[nd++] = (char)('0' + v % 10); while ((v /= 10) > 0);
                while (nd > 0) ctx += num[--nd];
                cond_list.push_back(ci);
            }
        }
        const int32_t cc = d->ctx_ids.find_or_add(ctx.data(), ctx.size(), d->ctxs);
        if ((size_t)cc == d->ctx_cond_list.size()) d->ctx_cond_list.push_back(cond_list);
        else d->ctx_cond_list[(size_t)cc] = cond_list; // contextConditionsList.put(cc, condList): same list by construction
        for (int32_t c : cond_list)
            if (c >= n_conds) {
                g_dao_err = "line " + std::to_string(ln + 1) + ": more condition columns than the header declares";
                delete d;
                return CMI_E_INVALID;
            }
        cells.push_back(Cell{((uint64_t)(uint32_t)uic << 32) | (uint32_t)cc, rate});
    }
    // ratingScale: sorted distinct values (DataDAO.java:348-350)
    std::sort(scale.begin(), scale.end());
    scale.erase(std::unique(scale.begin(), scale.end()), scale.end());
    d->rating_scale = scale;
    // CRS order; a stable sort keeps the file order inside a cell, whose LAST entry wins
    std::stable_sort(cells.begin(), cells.end(), [](const Cell &x, const Cell &y) { return x.key < y.key; });
    d->m_ui.reserve(cells.size());
    for (size_t i = 0; i < cells.size(); ++i) {
        if (i + 1 < cells.size() && cells[i + 1].key == cells[i].key) continue;
        d->m_ui.push_back((int32_t)(cells[i].key >> 32));
        d->m_ctx.push_back((int32_t)(cells[i].key & 0xffffffffu));
        d->m_r.push_back(cells[i].rate);
    }
    *out = d;
    return CMI_OK;
}

extern "C" int cmi_dao_read(const char *path, cmi_dao_handle *out) { return dao_read_impl(path, nullptr, out); }

extern "C" int cmi_dao_read_shared(const char *path, cmi_dao_handle train, cmi_dao_handle *out) {
    if (!train) {
        g_dao_err = "cmi_dao_read_shared: null training DAO";
        return CMI_E_INVALID;
    }
    return dao_read_impl(path, train, out);
}

extern "C" int cmi_dao_counts(cmi_dao_handle h, int64_t out[8]) {
    if (!h || !out) return CMI_E_INVALID;
    out[0] = (int64_t)h->users.size();
    out[1] = (int64_t)h->items.size();
    out[2] = (int64_t)h->ui_user.size();
    out[3] = (int64_t)h->ctxs.size();
    out[4] = (int64_t)h->conds.size();
    out[5] = (int64_t)h->dims.size();
    out[6] = h->num_ratings;
    out[7] = (int64_t)h->m_r.size();
    return CMI_OK;
}

extern "C" int cmi_dao_matrix(cmi_dao_handle h, int32_t *ui, int32_t *ctx, double *r) {
    if (!h) return CMI_E_INVALID;
    for (size_t t = 0; t < h->m_r.size(); ++t) {
        if (ui) ui[t] = h->m_ui[t];
        if (ctx) ctx[t] = h->m_ctx[t];
        if (r) r[t] = h->m_r[t];
    }
    return CMI_OK;
}

extern "C" int cmi_dao_ui_maps(cmi_dao_handle h, int32_t *ui_user, int32_t *ui_item) {
    if (!h) return CMI_E_INVALID;
    for (size_t t = 0; t < h->ui_user.size(); ++t) {
        if (ui_user) ui_user[t] = h->ui_user[t];
        if (ui_item) ui_item[t] = h->ui_item[t];
    }
    return CMI_OK;
}

extern "C" int64_t cmi_dao_ctx_nnz(cmi_dao_handle h) {
    if (!h) return 0;
    int64_t n = 0;
    for (auto &l : h->ctx_cond_list) n += (int64_t)l.size();
    return n;
}

extern "C" int cmi_dao_ctx_table(cmi_dao_handle h, int32_t *ctx_ptr, int32_t *ctx_conds) {
    if (!h || !ctx_ptr) return CMI_E_INVALID;
    int32_t off = 0;
    ctx_ptr[0] = 0;
    for (size_t c = 0; c < h->ctx_cond_list.size(); ++c) {
        for (int32_t v : h->ctx_cond_list[c])
            if (ctx_conds) ctx_conds[off++] = v;
            else off++;
        ctx_ptr[c + 1] = off;
    }
    return CMI_OK;
}

extern "C" int cmi_dao_cond_info(cmi_dao_handle h, int32_t *cond_dim, int32_t *empty_conds, int32_t *n_empty) {
    if (!h) return CMI_E_INVALID;
    if (cond_dim)
        for (size_t i = 0; i < h->cond_dim.size(); ++i) cond_dim[i] = h->cond_dim[i];
    if (empty_conds)
        for (size_t i = 0; i < h->empty_conds.size(); ++i) empty_conds[i] = h->empty_conds[i];
    if (n_empty) *n_empty = (int32_t)h->empty_conds.size();
    return CMI_OK;
}

extern "C" int cmi_dao_rating_scale(cmi_dao_handle h, double *out, int32_t cap, int32_t *n) {
    if (!h || !n) return CMI_E_INVALID;
    *n = (int32_t)h->rating_scale.size();
    if (out)
        for (int32_t i = 0; i < *n && i < cap; ++i) out[i] = h->rating_scale[(size_t)i];
    return CMI_OK;
}

// kind: 0 user, 1 item, 2 condition (header token), 3 context key ("c0,c1,.."), 4 dimension, 5 "u,i" pair key
extern "C" const char *cmi_dao_raw_id(cmi_dao_handle h, int kind, int32_t idx) {
    if (!h || idx < 0) return nullptr;
    const std::vector<std::string> *v = nullptr;
    switch (kind) {
    case 0: v = &h->users; break;
    case 1: v = &h->items; break;
    case 2: v = &h->conds; break;
    case 3: v = &h->ctxs; break;
    case 4: v = &h->dims; break;
    case 5: // the reference's key string of a (user, item) pair: inner ids joined by ',' (DataDAO.java:266)
        if ((size_t)idx >= h->ui_user.size()) return nullptr;
        h->scratch = std::to_string(h->ui_user[(size_t)idx]) + "," + std::to_string(h->ui_item[(size_t)idx]);
        return h->scratch.c_str();
    default: return nullptr;
    }
    return (size_t)idx < v->size() ? (*v)[(size_t)idx].c_str() : nullptr;
}

// ---- java.util.HashMap<String,?> iteration order ---------------------------------------------------------

namespace {

int32_t jstring_hash(const std::string &s) { // String.hashCode over UTF-16 code units (ASCII/Latin-1 bytes here)
    uint32_t h = 0;
    for (unsigned char c : s) h = 31u * h + c;
    return (int32_t)h;
}

// Keys in the order `for (K k : map.keySet())` visits them after inserting `keys` (first insertion of each
// distinct key) into a default-constructed HashMap: table 16, doubled whenever ++size > 0.75*capacity; bucket
// index (h ^ (h >>> 16)) & (cap-1); a resize splits every bin preserving relative order, so the final order is
// bucket index ascending, then insertion order.  (Bins that reach 8 entries while the table has >= 64 buckets
// are treeified and their root moves to the front; that needs >= 8 keys in one of >= 64 buckets and is not
// modelled -- the function reports whether any bin reached that size.)
std::vector<size_t> java_hashmap_order(const std::vector<std::string> &distinct_keys, bool *treeified) {
    size_t cap = 16;
    while ((double)distinct_keys.size() > 0.75 * (double)cap) cap <<= 1;
    std::vector<std::pair<uint32_t, size_t>> order;
    order.reserve(distinct_keys.size());
    std::vector<uint32_t> load(cap, 0);
    bool tree = false;
    for (size_t i = 0; i < distinct_keys.size(); ++i) {
        const uint32_t h = (uint32_t)jstring_hash(distinct_keys[i]);
        const uint32_t b = (h ^ (h >> 16)) & (uint32_t)(cap - 1);
        order.emplace_back(b, i);
        if (++load[b] >= 8 && cap >= 64) tree = true;
    }
    std::stable_sort(order.begin(), order.end(), [](const std::pair<uint32_t, size_t> &a, const std::pair<uint32_t, size_t> &b) { return a.first < b.first; });
    std::vector<size_t> out;
    out.reserve(order.size());
    for (auto &p : order) out.push_back(p.second);
    if (treeified) *treeified = tree;
    return out;
}

} // namespace

// positions[i] = index (into the caller's list of n keys, which must be distinct) of the i-th key visited
extern "C" int cmi_java_hashmap_order(int64_t n, const char *const *keys, int64_t *positions, int *treeified) {
    if (n < 0 || (n > 0 && (!keys || !positions))) return CMI_E_INVALID;
    std::vector<std::string> ks((size_t)n);
    for (int64_t i = 0; i < n; ++i) ks[(size_t)i] = keys[i];
    bool tree = false;
    const std::vector<size_t> ord = java_hashmap_order(ks, &tree);
    for (size_t i = 0; i < ord.size(); ++i) positions[i] = (int64_t)ord[i];
    if (treeified) *treeified = tree ? 1 : 0;
    return CMI_OK;
}

// ---- DataTransformer (src/carskit/data/processor/DataTransformer.java) -------------------------------------------

namespace {

// Multimap<String dim, String cond>: LinkedHashMultimap (insertion order of keys and of each key's values) when
// built while reading ONE file, TreeMultimap (both sorted by String.compareTo) when getConditions() merges the
// training and the test file (DataTransformer.java:57-92).
struct Conditions {
    bool sorted = false;
    std::vector<std::string> dims;
    std::map<std::string, std::vector<std::string>> conds;
    void put(const std::string &dim, const std::string &cond) {
        auto it = conds.find(dim);
        if (it == conds.end()) {
            it = conds.emplace(dim, std::vector<std::string>()).first;
            if (sorted) dims.insert(std::lower_bound(dims.begin(), dims.end(), dim), dim);
            else dims.push_back(dim);
        }
        std::vector<std::string> &v = it->second;
        if (sorted) {
            auto p = std::lower_bound(v.begin(), v.end(), cond);
            if (p == v.end() || *p != cond) v.insert(p, cond);
        } else if (std::find(v.begin(), v.end(), cond) == v.end()) {
            v.push_back(cond);
        }
    }
    bool has(const std::string &dim, const std::string &cond) const {
        auto it = conds.find(dim);
        return it != conds.end() && std::find(it->second.begin(), it->second.end(), cond) != it->second.end();
    }
};

// CARSKit.validateDataFormat (src/carskit/main/CARSKit.java:179-215): 1 binary, 2 loose, 3 compact; 0 where the reference THROWS
// (no data line: NullPointerException; a one-column header or a data line shorter than the header: ArrayIndexOutOfBounds; a value under
// a "dim:cond" column that Integer.valueOf refuses -- text, padding: NumberFormatException).  Statement order as in the reference: the
// `:` test short-circuits before the value is parsed, and isBinaryNumber (CARSKit.java:177) looks at the decimal digits with Java's
// truncating %, so every negative number passes (pinned against the interpreted source: tests/golden/reference_transform.json).
int validate_format(const std::vector<std::string> &lines) {
    if (lines.size() < 2) return 0;
    const std::vector<std::string> sh = split_keep(lines[0], ','), sd = split_keep(lines[1], ',');
    if (sh.size() < 2) return 0;
    if (jlower(jtrim(sh[sh.size() - 2])) == "dimension" && jlower(jtrim(sh.back())) == "condition") return 2;
    for (size_t i = 3; i < sh.size(); ++i) {
        if (sh[i].find(':') == std::string::npos) return 3;
        int32_t v = 0;
        if (i >= sd.size() || !jparse_int(sd[i], v)) return 0; // Integer.valueOf: no trim here
        for (int32_t c = v; c != 0; c /= 10)
            if (c % 10 > 1) return 3;
    }
    return 1;
}

typedef std::map<std::string, std::string> RatingContext; // HashMap<dim, cond> (only get() is used)

struct NewLines { // HashMap<String key, HashMap<dim,cond>> with first-insertion order remembered
    std::vector<std::string> keys;
    std::unordered_map<std::string, size_t> index;
    std::vector<RatingContext> ctx;
    RatingContext &at(const std::string &key, bool *fresh) {
        auto it = index.find(key);
        if (it == index.end()) {
            index.emplace(key, keys.size());
            keys.push_back(key);
            ctx.emplace_back();
            if (fresh) *fresh = true;
            return ctx.back();
        }
        if (fresh) *fresh = false;
        return ctx[it->second];
    }
};

// collectors used by getConditions() (DataTransformer.java:94-137)
bool collect_conditions(const std::vector<std::string> &lines, int fmt, Conditions &c, std::string &err) {
    const std::vector<std::string> header = split_keep(lines[0], ',');
    if (fmt == 1) {
        for (size_t i = 3; i < header.size(); ++i) {
            const std::vector<std::string> strs = split_keep(header[i], ':');
            if (strs.size() < 2) {
                err = "binary header token without ':'";
                return false;
            }
            c.put(jlower(jtrim(strs[0])), jlower(jtrim(strs[1])));
        }
    } else if (fmt == 2) {
        for (size_t ln = 1; ln < lines.size(); ++ln) {
            const std::vector<std::string> strs = split_keep(lines[ln], ',');
            if (strs.size() < 5) {
                err = "loose line with fewer than 5 fields";
                return false;
            }
            std::string cond = jlower(jtrim(strs[4]));
            if (cond.empty()) cond = "na";
            c.put(jlower(jtrim(strs[3])), cond);
        }
    } else {
        const size_t dimscount = header.size() - 3;
        for (size_t ln = 1; ln < lines.size(); ++ln) {
            const std::vector<std::string> strs = split_keep(lines[ln], ',');
            if (strs.size() < 3 + dimscount) {
                err = "compact line with fewer fields than the header";
                return false;
            }
            for (size_t i = 3; i < 3 + dimscount; ++i) {
                std::string cond = jlower(jtrim(strs[i]));
                if (cond.empty()) cond = "na";
                c.put(jlower(jtrim(header[i])), cond);
            }
        }
    }
    return true;
}

// One Transformation*ToBinary + PublishNewRatingFiles.  `given` != nullptr: the merged conditions of getConditions()
// (then a test file does not extend them); nullptr: a fresh LinkedHashMultimap filled while reading.
int transform_one(const std::vector<std::string> &lines, int fmt, bool is_test, const Conditions *given,
                  const char *out_path, bool *treeified) {
    Conditions own;
    Conditions &cond = own;
    if (given) own = *given;
    NewLines nl;
    const std::vector<std::string> header = split_keep(lines[0], ',');
    if (fmt == 3) { // DataTransformer.java:231-259
        if (header.size() < 3) {
            g_dao_err = "transform: header has fewer than 3 columns";
            return CMI_E_INVALID;
        }
        const size_t dimscount = header.size() - 3;
        std::vector<std::string> dims(dimscount);
        for (size_t i = 3; i < header.size(); ++i) dims[i - 3] = jlower(jtrim(header[i]));
        for (size_t ln = 1; ln < lines.size(); ++ln) {
            const std::vector<std::string> strs = split_keep(lines[ln], ',');
            if (strs.size() < 3 + dimscount) {
                g_dao_err = "transform: line " + std::to_string(ln + 1) + " has fewer fields than the header";
                return CMI_E_INVALID;
            }
            RatingContext rc;
            for (size_t i = 3; i < 3 + dimscount; ++i) {
                std::string c = jlower(jtrim(strs[i]));
                if (c.empty()) c = "na";
                rc[dims[i - 3]] = c;
                if (!is_test) cond.put(dims[i - 3], c);
            }
            nl.at(lines[ln], nullptr) = rc; // newlines.put(line, ratingcontext): the whole line is the key
        }
    } else if (fmt == 2) { // DataTransformer.java:196-229
        for (size_t ln = 1; ln < lines.size(); ++ln) {
            const std::vector<std::string> strs = split_keep(lines[ln], ',');
            if (strs.size() < 5) {
                g_dao_err = "transform: loose line " + std::to_string(ln + 1) + " has fewer than 5 fields";
                return CMI_E_INVALID;
            }
            const std::string key = jlower(jtrim(strs[0])) + "," + jlower(jtrim(strs[1])) + "," + jlower(jtrim(strs[2]));
            std::string c = jlower(jtrim(strs[4]));
            if (c.empty()) c = "na";
            const std::string dim = jlower(jtrim(strs[3]));
            if (!is_test) cond.put(dim, c);
            nl.at(key, nullptr)[dim] = c;
        }
    } else { // binary -> binary (DataTransformer.java:158-194)
        for (size_t ln = 1; ln < lines.size(); ++ln) {
            const std::vector<std::string> strs = split_keep(lines[ln], ',');
            if (strs.size() < header.size()) {
                g_dao_err = "transform: binary line " + std::to_string(ln + 1) + " has fewer fields than the header";
                return CMI_E_INVALID;
            }
            RatingContext rc;
            for (size_t i = 3; i < header.size(); ++i) {
                int32_t v;
                if (!jparse_int(jlower(jtrim(strs[i])), v)) {
                    g_dao_err = "transform: line " + std::to_string(ln + 1) + ": flag '" + strs[i] + "' is not an integer";
                    return CMI_E_INVALID;
                }
                if (v == 0) continue;
                const std::vector<std::string> rs = split_keep(header[i], ':');
                if (rs.size() < 2) {
                    g_dao_err = "transform: binary header token without ':'";
                    return CMI_E_INVALID;
                }
                rc[jlower(jtrim(rs[0]))] = jlower(jtrim(rs[1]));
                if (!is_test) cond.put(jlower(jtrim(rs[0])), jlower(jtrim(rs[1])));
            }
            nl.at(lines[ln], nullptr) = rc;
        }
    }
    // PublishNewRatingFiles (DataTransformer.java:266-329)
    bool tree = false;
    const std::vector<size_t> ord = java_hashmap_order(nl.keys, &tree);
    if (treeified) *treeified = *treeified || tree;
    FILE *f = fopen(out_path, "wb");
    if (!f) {
        g_dao_err = std::string("transform: cannot write ") + out_path;
        return CMI_E_INVALID;
    }
    std::string hd = "User, Item, Rating";
    for (const std::string &dim : cond.dims)
        for (const std::string &c : cond.conds[dim]) hd += ", " + dim + ":" + c;
    fprintf(f, "%s\n", hd.c_str());
    const bool is_loose = fmt == 2;
    for (size_t oi : ord) {
        const RatingContext &rc = nl.ctx[oi];
        std::string bits;
        for (const std::string &dim : cond.dims) {
            auto it = rc.find(dim);
            const bool missing = it == rc.end();
            if (missing && !is_loose) {
                fclose(f);
                g_dao_err = "transform: a rating has no condition for dimension '" + dim + "' (NullPointerException in the reference)";
                return CMI_E_INVALID;
            }
            const std::string dimCondition = missing ? std::string() : it->second;
            const bool isNA = missing || dimCondition == "na";
            bool isCompleted = false;
            for (const std::string &c : cond.conds[dim]) {
                if (!bits.empty()) bits += ",";
                if (is_loose) {
                    if (isNA) {
                        if (c == "na") {
                            bits += "1";
                            isCompleted = true;
                        } else bits += "0";
                    } else if (isCompleted) bits += "0";
                    else if (c == dimCondition) {
                        bits += "1";
                        isCompleted = true;
                    } else bits += "0";
                } else bits += (dimCondition == c) ? "1" : "0";
            }
        }
        std::string key = nl.keys[oi];
        const std::vector<std::string> skey = split_keep(key, ',');
        if (skey.size() > 3) key = jlower(jtrim(skey[0])) + "," + jlower(jtrim(skey[1])) + "," + jlower(jtrim(skey[2]));
        fprintf(f, "%s,%s\n", key.c_str(), bits.c_str());
    }
    fclose(f);
    return CMI_OK;
}

} // namespace

extern "C" int cmi_validate_data_format(const char *path) {
    std::vector<std::string> lines;
    if (!path || !read_lines(path, lines, g_dao_err)) return CMI_E_INVALID;
    return validate_format(lines);
}

// DataTransformer.run() (DataTransformer.java:331-396).  test_in == NULL: only the training file is converted
// (binary input is copied verbatim).  Otherwise getConditions() merges both files' conditions into a SORTED
// multimap (adding "na" to every dimension that lacks it) and both files are rewritten against it.
extern "C" int cmi_transform(const char *train_in, const char *train_out, const char *test_in, const char *test_out,
                             int *treeified) {
    if (treeified) *treeified = 0;
    if (!train_in || !train_out || (test_in && !test_out)) return CMI_E_INVALID;
    std::vector<std::string> tr, te;
    if (!read_lines(train_in, tr, g_dao_err)) return CMI_E_INVALID;
    const int ftr = validate_format(tr);
    if (ftr == 0) {
        g_dao_err = "transform: the training file is not a rating file validateDataFormat accepts (no data line, a short line, or a non-integer under a dim:cond column)";
        return CMI_E_INVALID;
    }
    bool tree = false;
    if (!test_in) {
        if (ftr == 1) { // FileIO.copyFile
            std::ifstream src(train_in, std::ios::binary);
            std::ofstream dst(train_out, std::ios::binary);
            dst << src.rdbuf();
            return dst ? CMI_OK : CMI_E_INVALID;
        }
        const int rc = transform_one(tr, ftr, false, nullptr, train_out, &tree);
        if (treeified) *treeified = tree;
        return rc;
    }
    if (!read_lines(test_in, te, g_dao_err)) return CMI_E_INVALID;
    const int fte = validate_format(te);
    if (fte == 0) {
        g_dao_err = "transform: the test file is not a rating file validateDataFormat accepts (no data line, a short line, or a non-integer under a dim:cond column)";
        return CMI_E_INVALID;
    }
    Conditions merged;
    merged.sorted = true;
    if (!collect_conditions(tr, ftr, merged, g_dao_err) || !collect_conditions(te, fte, merged, g_dao_err)) return CMI_E_INVALID;
    for (const std::string &dim : std::vector<std::string>(merged.dims))
        if (!merged.has(dim, "na")) merged.put(dim, "na");
    int rc = transform_one(tr, ftr, false, &merged, train_out, &tree);
    if (rc == CMI_OK) rc = transform_one(te, fte, true, &merged, test_out, &tree);
    if (treeified) *treeified = tree;
    return rc;
}

// kept for callers that know their input is compact
extern "C" int cmi_transform_compact_to_binary(const char *in_path, const char *out_path, int *treeified) {
    if (!in_path || !out_path) return CMI_E_INVALID;
    std::vector<std::string> lines;
    if (!read_lines(in_path, lines, g_dao_err)) return CMI_E_INVALID;
    if (lines.empty()) {
        g_dao_err = "transform: empty file";
        return CMI_E_INVALID;
    }
    bool tree = false;
    const int rc = transform_one(lines, 3, false, nullptr, out_path, &tree);
    if (treeified) *treeified = tree ? 1 : 0;
    return rc;
}

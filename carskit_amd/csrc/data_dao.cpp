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
    std::unordered_map<std::string, int32_t> user_ids, item_ids, ui_ids, ctx_ids, dim_ids;
    std::vector<std::string> users, items, uis, ctxs, dims, conds; // inner id -> raw key
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

extern "C" int cmi_dao_read(const char *path, cmi_dao_handle *out) {
    if (out) *out = nullptr;
    if (!path || !out) {
        g_dao_err = "cmi_dao_read: null argument";
        return CMI_E_INVALID;
    }
    std::vector<std::string> lines;
    if (!read_lines(path, lines, g_dao_err)) return CMI_E_INVALID;
    if (lines.empty()) {
        g_dao_err = "cmi_dao_read: empty file (the reference dereferences a null header line)";
        return CMI_E_INVALID;
    }
    cmi_dao *d = new cmi_dao();
    // header (DataDAO.java:198-215): trim, split on runs of tab/comma, columns >= 3 are conditions
    {
        const std::vector<std::string> hd = split_runs(jtrim(lines[0]));
        for (size_t i = 3; i < hd.size(); ++i) {
            const std::string context = jtrim(hd[i]);
            const size_t colon = context.find(':');
            // context.split(":")[0]: text before the first ':' (an empty first token stays empty)
            const std::string dim = jtrim(colon == std::string::npos ? context : context.substr(0, colon));
            const int32_t dimc = first_seen(d->dim_ids, d->dims, dim);
            d->conds.push_back(context);
            d->cond_dim.push_back(dimc);
            const std::string na = ":na";
            if (context.size() >= na.size() && context.compare(context.size() - na.size(), na.size(), na) == 0)
                d->empty_conds.push_back((int32_t)i - 3);
        }
    }
    const int32_t n_conds = (int32_t)d->conds.size();
    // data lines (DataDAO.java:222-345); dataTable.put(uic, cc, rate): the LAST line of a (ui, ctx) cell wins
    std::map<std::pair<int32_t, int32_t>, double> table; // ordered by (ui, ctx) = CRS order
    std::vector<double> scale;
    for (size_t ln = 1; ln < lines.size(); ++ln) {
        const std::vector<std::string> data = split_keep(jtrim(lines[ln]), ',');
        if (data.size() < 3) {
            d->err = "line " + std::to_string(ln + 1) + ": fewer than 3 fields (ArrayIndexOutOfBounds in the reference)";
            g_dao_err = d->err;
            delete d;
            return CMI_E_INVALID;
        }
        double rate;
        if (!jparse_double(data[2], rate)) {
            g_dao_err = "line " + std::to_string(ln + 1) + ": rating '" + data[2] + "' is not a number (NumberFormatException)";
            delete d;
            return CMI_E_INVALID;
        }
        scale.push_back(rate);
        d->num_ratings++;
        const int32_t row = first_seen(d->user_ids, d->users, data[0]); // NOT trimmed (DataDAO.java:226-227)
        const int32_t col = first_seen(d->item_ids, d->items, data[1]);
        const std::string useritem = std::to_string(row) + "," + std::to_string(col);
        const int32_t uic = first_seen(d->ui_ids, d->uis, useritem);
        if ((size_t)uic == d->ui_user.size()) {
            d->ui_user.push_back(row);
            d->ui_item.push_back(col);
        }
        std::string ctx;
        std::vector<int32_t> cond_list;
        for (size_t i = 3; i < data.size(); ++i) {
            int32_t value;
            if (!jparse_int(jtrim(data[i]), value)) {
                g_dao_err = "line " + std::to_string(ln + 1) + ": condition flag '" + data[i] + "' is not an integer (NumberFormatException)";
                delete d;
                return CMI_E_INVALID;
            }
            if (value == 1) {
                if (!ctx.empty()) ctx += ",";
                ctx += std::to_string(i - 3);
                cond_list.push_back((int32_t)i - 3);
            }
        }
        const int32_t cc = first_seen(d->ctx_ids, d->ctxs, ctx);
        if ((size_t)cc == d->ctx_cond_list.size()) d->ctx_cond_list.push_back(cond_list);
        else d->ctx_cond_list[(size_t)cc] = cond_list; // contextConditionsList.put(cc, condList): same list by construction
        for (int32_t c : cond_list)
            if (c >= n_conds) {
                g_dao_err = "line " + std::to_string(ln + 1) + ": more condition columns than the header declares";
                delete d;
                return CMI_E_INVALID;
            }
        table[{uic, cc}] = rate;
    }
    // ratingScale: sorted distinct values (DataDAO.java:348-350)
    std::sort(scale.begin(), scale.end());
    scale.erase(std::unique(scale.begin(), scale.end()), scale.end());
    d->rating_scale = scale;
    d->m_ui.reserve(table.size());
    for (auto &kv : table) {
        d->m_ui.push_back(kv.first.first);
        d->m_ctx.push_back(kv.first.second);
        d->m_r.push_back(kv.second);
    }
    *out = d;
    return CMI_OK;
}

extern "C" int cmi_dao_counts(cmi_dao_handle h, int64_t out[8]) {
    if (!h || !out) return CMI_E_INVALID;
    out[0] = (int64_t)h->users.size();
    out[1] = (int64_t)h->items.size();
    out[2] = (int64_t)h->uis.size();
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
    case 5: v = &h->uis; break;
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

// DataTransformer.TransformationFromCompactToBinary + PublishNewRatingFiles (isLoose=false) + getHeader.
// *treeified (may be NULL) is set if a HashMap bin reached the treeify threshold (row order then not guaranteed).
extern "C" int cmi_transform_compact_to_binary(const char *in_path, const char *out_path, int *treeified) {
    if (!in_path || !out_path) return CMI_E_INVALID;
    std::vector<std::string> lines;
    if (!read_lines(in_path, lines, g_dao_err)) return CMI_E_INVALID;
    if (lines.empty()) {
        g_dao_err = "transform: empty file";
        return CMI_E_INVALID;
    }
    const std::vector<std::string> header = split_keep(lines[0], ','); // NOT trimmed as a whole (DataTransformer.java:234)
    if (header.size() < 3) {
        g_dao_err = "transform: header has fewer than 3 columns";
        return CMI_E_INVALID;
    }
    const size_t dimscount = header.size() - 3;
    std::vector<std::string> dims(dimscount);
    for (size_t i = 3; i < header.size(); ++i) dims[i - 3] = jlower(jtrim(header[i]));
    // conditions: LinkedHashMultimap -> dims in first-put order, conditions of a dim in first-put order, no duplicates
    std::vector<std::string> dim_order;
    std::unordered_map<std::string, std::vector<std::string>> dim_conds;
    // newlines: HashMap<line, HashMap<dim, cond>>; a repeated line overwrites (same value anyway)
    std::vector<std::string> keys;
    std::unordered_map<std::string, size_t> key_index;
    std::vector<std::vector<std::string>> key_conds; // per distinct line: condition of dims[d]
    for (size_t ln = 1; ln < lines.size(); ++ln) {
        const std::vector<std::string> strs = split_keep(lines[ln], ',');
        if (strs.size() < 3 + dimscount) {
            g_dao_err = "transform: line " + std::to_string(ln + 1) + " has fewer fields than the header";
            return CMI_E_INVALID;
        }
        std::vector<std::string> rc(dimscount);
        // ratingcontext.put(dims[d], cond): a later column with the same dim name overwrites an earlier one
        std::unordered_map<std::string, std::string> by_dim;
        for (size_t i = 3; i < 3 + dimscount; ++i) {
            std::string cond = jlower(jtrim(strs[i]));
            if (cond.empty()) cond = "na";
            by_dim[dims[i - 3]] = cond;
            auto it = dim_conds.find(dims[i - 3]);
            if (it == dim_conds.end()) {
                dim_order.push_back(dims[i - 3]);
                it = dim_conds.emplace(dims[i - 3], std::vector<std::string>()).first;
            }
            if (std::find(it->second.begin(), it->second.end(), cond) == it->second.end()) it->second.push_back(cond);
        }
        for (size_t dd = 0; dd < dimscount; ++dd) rc[dd] = by_dim[dims[dd]];
        auto ki = key_index.find(lines[ln]);
        if (ki == key_index.end()) {
            key_index.emplace(lines[ln], keys.size());
            keys.push_back(lines[ln]);
            key_conds.push_back(rc);
        } else {
            key_conds[ki->second] = rc;
        }
    }
    bool tree = false;
    const std::vector<size_t> ord = java_hashmap_order(keys, &tree);
    if (treeified) *treeified = tree ? 1 : 0;
    FILE *f = fopen(out_path, "wb");
    if (!f) {
        g_dao_err = std::string("transform: cannot write ") + out_path;
        return CMI_E_INVALID;
    }
    std::string hd = "User, Item, Rating";
    for (const std::string &dim : dim_order)
        for (const std::string &cond : dim_conds[dim]) hd += ", " + dim + ":" + cond;
    fprintf(f, "%s\n", hd.c_str());
    for (size_t oi : ord) {
        std::string bits;
        for (const std::string &dim : dim_order) {
            // ratingcontext.get(dim): the line's condition for this dimension name
            std::string dimCondition;
            for (size_t dd = 0; dd < dimscount; ++dd)
                if (dims[dd] == dim) dimCondition = key_conds[oi][dd];
            for (const std::string &cond : dim_conds[dim]) {
                if (!bits.empty()) bits += ",";
                bits += (dimCondition == cond) ? "1" : "0";
            }
        }
        std::string key = keys[oi];
        const std::vector<std::string> skey = split_keep(key, ',');
        if (skey.size() > 3) key = jlower(jtrim(skey[0])) + "," + jlower(jtrim(skey[1])) + "," + jlower(jtrim(skey[2]));
        fprintf(f, "%s,%s\n", key.c_str(), bits.c_str());
    }
    fclose(f);
    return CMI_OK;
}

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
#include <chrono>
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

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "host_pool.hpp"

using cmi::host_threads;
using cmi::parallel_ranges;

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

// The bytes of a rating file: mapped (large files: no copy, the page cache's pages) or read into a buffer.
class FileImage {
  public:
    FileImage() = default;
    FileImage(const FileImage &) = delete;
    FileImage &operator=(const FileImage &) = delete;
    ~FileImage() {
        if (map_) munmap((void *)p_, n_);
    }
    bool open(const char *path, std::string &err) {
        const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
        if (fd < 0) {
            err = std::string("cannot open ") + path;
            return false;
        }
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= ((off_t)16 << 20)) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                p_ = (const char *)m;
                n_ = (size_t)st.st_size;
                map_ = true;
                ::close(fd);
                return true;
            }
        }
        std::string buf;
        char tmp[1 << 16];
        for (;;) {
            const ssize_t got = ::read(fd, tmp, sizeof tmp);
            if (got <= 0) break;
            buf.append(tmp, (size_t)got);
        }
        ::close(fd);
        own_.swap(buf);
        p_ = own_.data();
        n_ = own_.size();
        return true;
    }
    const char *data() const { return p_; }
    size_t size() const { return n_; }
    char operator[](size_t i) const { return p_[i]; }
    std::string substr(size_t b, size_t len) const { return std::string(p_ + b, std::min(len, n_ - b)); }

  private:
    const char *p_ = nullptr;
    size_t n_ = 0;
    bool map_ = false;
    std::string own_;
};

// the same line structure without one std::string per line: (offset, length) spans into the file image
bool read_spans(const char *path, FileImage &all, std::vector<std::pair<size_t, size_t>> &spans, std::string &err) {
    if (!all.open(path, err)) return false;
    size_t i = 0, b = 0;
    const size_t n = all.size();
    size_t par_min = (size_t)16 << 20;
    if (const char *e = getenv("CMI_DAO_PARALLEL_MIN_LINES")) par_min = (size_t)std::max(1ll, atoll(e)); // tests: small files through this path
    const int nt = cmi::host_threads((int64_t)n / 64);
    if (n >= par_min && nt > 1) {
        // Large files: the line breaks are found in ranges of BYTES on the host's cores.  A break starts at every '\r' and at every '\n'
        // that does not follow a '\r' (that one belongs to the break the '\r' started): a property of a byte and its predecessor, so
        // the ranges need nothing from each other; a line's start is where the previous break ended.
        struct Brk {
            size_t at, next; // first byte of the break; first byte after it
        };
        std::vector<std::vector<Brk>> found((size_t)nt);
        cmi::parallel_ranges((int64_t)n, nt, [&](int part, int64_t lo, int64_t hi) {
            std::vector<Brk> &v = found[(size_t)part];
            v.reserve((size_t)(hi - lo) / 32 + 16);
            for (size_t x = (size_t)lo; x < (size_t)hi; ++x) {
                const char c = all[x];
                if (c == '\r') v.push_back(Brk{x, x + 1 < n && all[x + 1] == '\n' ? x + 2 : x + 1});
                else if (c == '\n' && !(x > 0 && all[x - 1] == '\r')) v.push_back(Brk{x, x + 1});
            }
        });
        std::vector<size_t> off((size_t)nt + 1, 0);
        for (int t = 0; t < nt; ++t) off[(size_t)t + 1] = off[(size_t)t] + found[(size_t)t].size();
        const size_t nb = off[(size_t)nt];
        size_t last_next = 0;
        for (int t = nt - 1; t >= 0 && last_next == 0; --t)
            if (!found[(size_t)t].empty()) last_next = found[(size_t)t].back().next;
        spans.resize(nb + (last_next < n ? 1 : 0));
        cmi::parallel_ranges(nt, nt, [&](int, int64_t t0, int64_t t1) {
            for (int64_t t = t0; t < t1; ++t) {
                size_t start = 0; // where the line of this range's first break begins: the previous break's end
                for (int64_t q = t - 1; q >= 0; --q)
                    if (!found[(size_t)q].empty()) {
                        start = found[(size_t)q].back().next;
                        break;
                    }
                size_t k = off[(size_t)t];
                for (const Brk &br : found[(size_t)t]) {
                    spans[k++] = std::make_pair(start, br.at - start);
                    start = br.next;
                }
            }
        });
        if (last_next < n) spans[nb] = std::make_pair(last_next, n - last_next);
        return true;
    }
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
    int32_t find(const char *p, size_t n, const std::vector<std::string> &names) const { // -1: not present
        if (slots_.empty()) return -1;
        const uint64_t h = hash(p, n);
        size_t i = (size_t)h & (slots_.size() - 1);
        while (slots_[i].id >= 0) {
            if (slots_[i].h == h) {
                const std::string &s = names[(size_t)slots_[i].id];
                if (s.size() == n && std::memcmp(s.data(), p, n) == 0) return slots_[i].id;
            }
            i = (i + 1) & (slots_.size() - 1);
        }
        return -1;
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

// (user, item) pair -> ui id, first seen (the reference keys a HashMap by the string "<u>,<i>" of inner ids).  SHARDS independent open-
// addressing tables selected by the key's hash: the ranged reader fills them side by side (one thread per shard), the sequential one
// through find_or_add as before.
class PairIndex {
  public:
    static constexpr int SHARDS = 16;
    static int shard_of(uint64_t key) { return (int)(mix(key) >> 60); }
    // returns the id; *fresh tells whether the pair was new (then id == next_id)
    int32_t find_or_add(uint64_t key, int32_t next_id, bool *fresh) { return sub_[shard_of(key)].find_or_add(key, next_id, fresh); }
    int32_t find(uint64_t key) const { return sub_[shard_of(key)].find(key); } // -1: not present
    // one shard's table, for a thread that owns that shard
    int32_t shard_find_or_add(int s, uint64_t key, int32_t id, bool *fresh) { return sub_[s].find_or_add(key, id, fresh); }
    void shard_set(int s, uint64_t key, int32_t id) { sub_[s].set(key, id); }

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
    struct Sub {
        int32_t find_or_add(uint64_t key, int32_t next_id, bool *fresh) {
            if ((size_ + 1) * 2 > slots_.size()) grow();
            size_t i = (size_t)mix(key) & (slots_.size() - 1);
            while (slots_[i].id != EMPTY) {
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
        int32_t find(uint64_t key) const {
            if (slots_.empty()) return -1;
            size_t i = (size_t)mix(key) & (slots_.size() - 1);
            while (slots_[i].id != EMPTY) {
                if (slots_[i].key == key) return slots_[i].id;
                i = (i + 1) & (slots_.size() - 1);
            }
            return -1;
        }
        void set(uint64_t key, int32_t id) { // the key is present
            size_t i = (size_t)mix(key) & (slots_.size() - 1);
            while (slots_[i].key != key || slots_[i].id == EMPTY) i = (i + 1) & (slots_.size() - 1);
            slots_[i].id = id;
        }
        void grow() {
            std::vector<Slot> old;
            old.swap(slots_);
            slots_.assign(old.empty() ? 256 : old.size() * 2, Slot{0, EMPTY});
            for (const Slot &sl : old)
                if (sl.id != EMPTY) {
                    size_t i = (size_t)mix(sl.key) & (slots_.size() - 1);
                    while (slots_[i].id != EMPTY) i = (i + 1) & (slots_.size() - 1);
                    slots_[i] = sl;
                }
        }
        static constexpr int32_t EMPTY = INT32_MIN; // (ids are >= 0; the ranged reader parks a negative placeholder for a moment)
        std::vector<Slot> slots_;
        size_t size_ = 0;
    };
    Sub sub_[SHARDS];
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

static int dao_read_body(const char *path, const cmi_dao *base, cmi_dao_handle *out);
static int dao_read_impl(const char *path, const cmi_dao *base, cmi_dao_handle *out) { // exception barrier (ranged reader on the host pool)
    try {
        return dao_read_body(path, base, out);
    } catch (const std::exception &e) {
        if (out) *out = nullptr;
        g_dao_err = std::string("cmi_dao_read: host-side failure: ") + e.what();
        return CMI_E_HOST;
    } catch (...) {
        if (out) *out = nullptr;
        g_dao_err = "cmi_dao_read: host-side failure (unknown exception)";
        return CMI_E_HOST;
    }
}
static int dao_read_body(const char *path, const cmi_dao *base, cmi_dao_handle *out) {
    if (out) *out = nullptr;
    if (!path || !out) {
        g_dao_err = "cmi_dao_read: null argument";
        return CMI_E_INVALID;
    }
    const bool times = getenv("CMI_DAO_TIMES") != nullptr; // tools/exp/dao_read_time.py
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *w) {
        if (!times) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "dao %s %.3f s\n", w, std::chrono::duration<double>(n - T0).count());
        T0 = n;
    };
    FileImage image;
    std::vector<std::pair<size_t, size_t>> lines; // spans into `image`
    if (!read_spans(path, image, lines, g_dao_err)) return CMI_E_INVALID;
    lap("read+split");
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
    std::vector<double> scale;
    // One line -> its three raw keys and its rating.  `ctx` receives the context key (the comma-joined indices of the columns that
    // hold 1), `cond_list` the indices.  Returns false with `err` set for what the reference would throw on.
    auto parse_line = [&image, &lines, n_conds](size_t ln, const char *&ub, const char *&ue, const char *&ib, const char *&ie, double &rate,
                                                std::string &ctx, std::vector<int32_t> &cond_list, std::string &err) -> bool {
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
        const char *rb, *re;
        if (!next_field(ub, ue) || !next_field(ib, ie) || !next_field(rb, re)) {
            err = "line " + std::to_string(ln + 1) + ": fewer than 3 fields (ArrayIndexOutOfBounds in the reference)";
            return false;
        }
        rate = 0.0;
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
                err = "line " + std::to_string(ln + 1) + ": rating '" + std::string(rb, re) + "' is not a number (NumberFormatException)";
                return false;
            }
        }
        ctx.clear();
        cond_list.clear();
        auto add_cond = [&](int32_t ci) {
            if (!ctx.empty()) ctx += ',';
            char num[12];
            int nd = 0, v = ci;
            do num[nd++] = (char)('0' + v % 10); while ((v /= 10) > 0);
            while (nd > 0) ctx += num[--nd];
            cond_list.push_back(ci);
        };
        // the usual shape of the flag columns -- single characters 0 / 1 between commas -- is read byte by byte; anything else (spaces,
        // other integers, empty fields) goes through the general field scan below from the start of the flags
        bool plain = p <= end;
        if (plain) {
            const char *q = p;
            int32_t ci = 0;
            for (; q < end; q += 2, ++ci) {
                if ((*q != '0' && *q != '1') || (q + 1 < end && q[1] != ',')) {
                    plain = false;
                    break;
                }
                if (q + 1 == end) { // the last flag
                    if (*q == '1') add_cond(ci);
                    q = end + 1;
                    break;
                }
                if (*q == '1') add_cond(ci);
            }
            if (plain && q == end) plain = false; // an empty line tail or a trailing comma (an empty last field): the general scan
            if (plain) p = end + 1;
            else {
                ctx.clear();
                cond_list.clear();
            }
        }
        const char *fb, *fe;
        for (int32_t ci = 0; next_field(fb, fe); ++ci) {
            int32_t value;
            while (fb < fe && (unsigned char)*fb <= ' ') ++fb; // data[i].trim()
            while (fe > fb && (unsigned char)fe[-1] <= ' ') --fe;
            if (fe - fb == 1 && (*fb == '0' || *fb == '1')) value = *fb - '0';
            else if (!jparse_int(std::string(fb, fe), value)) {
                err = "line " + std::to_string(ln + 1) + ": condition flag '" + std::string(fb, fe) + "' is not an integer (NumberFormatException)";
                return false;
            }
            if (value == 1) add_cond(ci);
        }
        for (int32_t c : cond_list)
            if (c >= n_conds) {
                err = "line " + std::to_string(ln + 1) + ": more condition columns than the header declares";
                return false;
            }
        return true;
    };

    // Large files: the lines are parsed in RANGES on the host's cores.  Ids are first-seen ranks, and a key's first occurrence lies in
    // the earliest range that holds it: every range interns the keys it does not find in the tables it started with into tables of its
    // own (first-seen order inside the range), the ranges' new keys are then entered into the shared tables range after range -- the
    // order the sequential pass meets them -- and the lines are translated.  The same twice over for the (user, item) pairs, whose
    // keys are made of the users' and items' final ids.  Any error: the sequential pass below runs instead and reports it.
    size_t par_min = (size_t)1 << 16;
    if (const char *e = getenv("CMI_DAO_PARALLEL_MIN_LINES")) par_min = (size_t)std::max(1ll, atoll(e)); // tests: small files through this path
    const int64_t n_lines = (int64_t)lines.size() - 1;
    const int nt = host_threads(n_lines);
    bool parallel_done = false;
    if ((size_t)n_lines >= par_min && nt > 1) {
        struct Range {
            StrIndex users_i, items_i, ctxs_i;
            std::vector<std::string> users, items, ctxs;       // keys new to this range, first-seen
            std::vector<std::vector<int32_t>> ctx_conds;        // per new context: its condition list
            PairIndex pairs_i;
            std::vector<uint64_t> pairs;                        // (user, item) pairs new to this range, first-seen
            std::vector<int32_t> by_shard[PairIndex::SHARDS];   // their positions in `pairs`, per shard of the pair table
            std::vector<uint8_t> fresh;                         // pair k is new to the whole file so far
            std::vector<int32_t> rank;                          // number of fresh pairs before k in this range
            int64_t first_id = 0;                               // ui id of the range's first fresh pair
            std::vector<int32_t> tu, ti, tc;                    // new key -> final id
            int64_t b = 0, e = 0;
            bool bad = false;
        };
        std::vector<Range> rg((size_t)nt);
        // per line: >= 0 = the range's own new key, < 0 = ~(id in the tables the read started with)
        std::unique_ptr<int32_t[]> lu(new int32_t[(size_t)n_lines]), li(new int32_t[(size_t)n_lines]), lc(new int32_t[(size_t)n_lines]);
        std::unique_ptr<double[]> lr(new double[(size_t)n_lines]);
        const cmi_dao *dc = d; // read-only while the ranges run
        parallel_ranges(n_lines, nt, [&](int part, int64_t b, int64_t e) {
            Range &R = rg[(size_t)part];
            R.b = b;
            R.e = e;
            std::string ctx, err;
            std::vector<int32_t> cond_list;
            for (int64_t x = b; x < e; ++x) {
                const char *ub, *ue, *ib, *ie;
                if (!parse_line((size_t)x + 1, ub, ue, ib, ie, lr[(size_t)x], ctx, cond_list, err)) {
                    R.bad = true;
                    return;
                }
                int32_t g = dc->user_ids.find(ub, (size_t)(ue - ub), dc->users); // NOT trimmed (DataDAO.java:226-227)
                lu[(size_t)x] = g >= 0 ? ~g : R.users_i.find_or_add(ub, (size_t)(ue - ub), R.users);
                g = dc->item_ids.find(ib, (size_t)(ie - ib), dc->items);
                li[(size_t)x] = g >= 0 ? ~g : R.items_i.find_or_add(ib, (size_t)(ie - ib), R.items);
                g = dc->ctx_ids.find(ctx.data(), ctx.size(), dc->ctxs);
                if (g >= 0) lc[(size_t)x] = ~g;
                else {
                    const int32_t l = R.ctxs_i.find_or_add(ctx.data(), ctx.size(), R.ctxs);
                    if ((size_t)l == R.ctx_conds.size()) R.ctx_conds.push_back(cond_list);
                    lc[(size_t)x] = l;
                }
            }
        });
        bool bad = false;
        for (const Range &R : rg) bad = bad || R.bad;
        lap("parse");
        if (!bad) {
            // The ranges' new keys enter the shared tables range after range.  For the users and the items (millions of keys, and with
            // unordered data every range meets most of them) that order is produced by a TREE of pairwise merges first: merging table B
            // into table A appends B's keys that A does not hold, in B's order -- an associative step, so ranges (0,1), (2,3), ... merge
            // side by side, then (01, 23), ... and range 0's table ends up holding every new key in the sequential pass's order; those
            // enter the shared table once, and every range looks its own keys up (in parallel).
            {
                std::vector<size_t> nu0((size_t)nt), ni0((size_t)nt);
                for (int r = 0; r < nt; ++r) {
                    nu0[(size_t)r] = rg[(size_t)r].users.size();
                    ni0[(size_t)r] = rg[(size_t)r].items.size();
                }
                for (int stride = 1; stride < nt; stride *= 2) {
                    const int64_t pairs = (nt + 2 * stride - 1) / (2 * stride);
                    parallel_ranges(pairs, (int)std::min<int64_t>(pairs, nt), [&](int, int64_t q0, int64_t q1) {
                        for (int64_t q = q0; q < q1; ++q) {
                            const int64_t a = q * 2 * stride, b = a + stride;
                            if (b >= nt) continue;
                            Range &A = rg[(size_t)a];
                            const Range &B = rg[(size_t)b];
                            for (const std::string &k : B.users) A.users_i.find_or_add(k.data(), k.size(), A.users);
                            for (const std::string &k : B.items) A.items_i.find_or_add(k.data(), k.size(), A.items);
                        }
                    });
                }
                for (const std::string &k : rg[0].users) d->user_ids.find_or_add(k.data(), k.size(), d->users);
                for (const std::string &k : rg[0].items) d->item_ids.find_or_add(k.data(), k.size(), d->items);
                parallel_ranges(nt, nt, [&](int, int64_t p0, int64_t p1) {
                    for (int64_t p = p0; p < p1; ++p) {
                        Range &R = rg[(size_t)p];
                        R.tu.resize(nu0[(size_t)p]);
                        R.ti.resize(ni0[(size_t)p]);
                        for (size_t k = 0; k < R.tu.size(); ++k) R.tu[k] = dc->user_ids.find(R.users[k].data(), R.users[k].size(), dc->users);
                        for (size_t k = 0; k < R.ti.size(); ++k) R.ti[k] = dc->item_ids.find(R.items[k].data(), R.items[k].size(), dc->items);
                    }
                });
            }
            for (Range &R : rg) { // the contexts (few): range after range
                R.tc.resize(R.ctxs.size());
                for (size_t k = 0; k < R.ctxs.size(); ++k) {
                    const int32_t cc = d->ctx_ids.find_or_add(R.ctxs[k].data(), R.ctxs[k].size(), d->ctxs);
                    R.tc[k] = cc;
                    if ((size_t)cc == d->ctx_cond_list.size()) d->ctx_cond_list.push_back(R.ctx_conds[k]);
                    else d->ctx_cond_list[(size_t)cc] = R.ctx_conds[k]; // (the same list by construction)
                }
            }
            lap("merge keys");
            // final user / item / context ids per line; the pairs new to each range
            parallel_ranges(nt, nt, [&](int, int64_t p0, int64_t p1) {
                for (int64_t p = p0; p < p1; ++p) {
                    Range &R = rg[(size_t)p];
                    for (int64_t x = R.b; x < R.e; ++x) {
                        const int32_t row = lu[(size_t)x] < 0 ? ~lu[(size_t)x] : R.tu[(size_t)lu[(size_t)x]];
                        const int32_t col = li[(size_t)x] < 0 ? ~li[(size_t)x] : R.ti[(size_t)li[(size_t)x]];
                        lc[(size_t)x] = lc[(size_t)x] < 0 ? ~lc[(size_t)x] : R.tc[(size_t)lc[(size_t)x]];
                        lu[(size_t)x] = row;
                        li[(size_t)x] = col;
                        const uint64_t uikey = ((uint64_t)(uint32_t)row << 32) | (uint32_t)col;
                        if (dc->ui_ids.find(uikey) < 0) {
                            bool fresh = false;
                            R.pairs_i.find_or_add(uikey, (int32_t)R.pairs.size(), &fresh);
                            if (fresh) {
                                R.by_shard[PairIndex::shard_of(uikey)].push_back((int32_t)R.pairs.size());
                                R.pairs.push_back(uikey);
                            }
                        }
                    }
                }
            });
            lap("translate");
            // The ranges' pairs enter the pair table range after range, as the keys did -- but there is one pair per line, so this
            // merge is done per SHARD of the table, all shards at once: a shard's thread walks the ranges in order and enters its
            // shard's pairs (placeholder ids), noting which were new to the file; the ids are then the ranks of the new pairs in
            // (range, first-seen) order -- a count per range, a scan over the ranges -- and a second sharded pass stores them.
            for (Range &R : rg) R.fresh.assign(R.pairs.size(), 0);
            parallel_ranges(PairIndex::SHARDS, std::min(nt, (int)PairIndex::SHARDS), [&](int, int64_t s0, int64_t s1) {
                for (int64_t sh = s0; sh < s1; ++sh)
                    for (Range &R : rg)
                        for (int32_t k : R.by_shard[sh]) {
                            bool fresh = false;
                            d->ui_ids.shard_find_or_add((int)sh, R.pairs[(size_t)k], -1, &fresh);
                            R.fresh[(size_t)k] = fresh ? 1 : 0;
                        }
            });
            parallel_ranges(nt, nt, [&](int, int64_t p0, int64_t p1) {
                for (int64_t p = p0; p < p1; ++p) {
                    Range &R = rg[(size_t)p];
                    R.rank.resize(R.pairs.size());
                    int32_t run = 0;
                    for (size_t k = 0; k < R.pairs.size(); ++k) {
                        R.rank[k] = run;
                        run += R.fresh[k];
                    }
                    R.first_id = run; // (count, turned into the first id below)
                }
            });
            int64_t next_ui = (int64_t)d->ui_user.size();
            for (Range &R : rg) {
                const int64_t cnt = R.first_id;
                R.first_id = next_ui;
                next_ui += cnt;
            }
            d->ui_user.resize((size_t)next_ui);
            d->ui_item.resize((size_t)next_ui);
            parallel_ranges(PairIndex::SHARDS, std::min(nt, (int)PairIndex::SHARDS), [&](int, int64_t s0, int64_t s1) {
                for (int64_t sh = s0; sh < s1; ++sh)
                    for (Range &R : rg)
                        for (int32_t k : R.by_shard[sh])
                            if (R.fresh[(size_t)k]) {
                                const uint64_t uikey = R.pairs[(size_t)k];
                                const int64_t id = R.first_id + R.rank[(size_t)k];
                                d->ui_ids.shard_set((int)sh, uikey, (int32_t)id);
                                d->ui_user[(size_t)id] = (int32_t)(uikey >> 32);
                                d->ui_item[(size_t)id] = (int32_t)(uikey & 0xffffffffu);
                            }
            });
            lap("merge pairs");
            cells.resize((size_t)n_lines);
            parallel_ranges(n_lines, nt, [&](int, int64_t b, int64_t e) {
                for (int64_t x = b; x < e; ++x) {
                    const uint64_t uikey = ((uint64_t)(uint32_t)lu[(size_t)x] << 32) | (uint32_t)li[(size_t)x];
                    cells[(size_t)x] = Cell{((uint64_t)(uint32_t)dc->ui_ids.find(uikey) << 32) | (uint32_t)lc[(size_t)x], lr[(size_t)x]};
                }
            });
            // ratingScale candidates: the distinct values of every range
            std::vector<std::vector<double>> sc((size_t)nt);
            parallel_ranges(nt, nt, [&](int, int64_t p0, int64_t p1) {
                for (int64_t p = p0; p < p1; ++p) {
                    std::vector<double> &v = sc[(size_t)p];
                    v.assign(lr.get() + rg[(size_t)p].b, lr.get() + rg[(size_t)p].e);
                    std::sort(v.begin(), v.end());
                    v.erase(std::unique(v.begin(), v.end()), v.end());
                }
            });
            for (const std::vector<double> &v : sc) scale.insert(scale.end(), v.begin(), v.end());
            d->num_ratings += n_lines;
            lap("cells+scale");
            // CRS order: every range sorted on its own (stable), then merged pairwise (std::merge takes from the earlier range on ties:
            // the file order inside a cell survives)
            std::vector<int64_t> cut;
            for (const Range &R : rg) cut.push_back(R.b);
            cut.push_back(n_lines);
            parallel_ranges(nt, nt, [&](int, int64_t p0, int64_t p1) {
                for (int64_t p = p0; p < p1; ++p)
                    std::stable_sort(cells.begin() + cut[(size_t)p], cells.begin() + cut[(size_t)p + 1], [](const Cell &x, const Cell &y) { return x.key < y.key; });
            });
            std::vector<Cell> tmp(cells.size());
            while (cut.size() > 2) {
                const int64_t pairs = (int64_t)(cut.size() - 1) / 2;
                parallel_ranges(pairs, (int)std::min<int64_t>(pairs, nt), [&](int, int64_t q0, int64_t q1) {
                    for (int64_t q = q0; q < q1; ++q) {
                        const int64_t a = cut[(size_t)(2 * q)], m = cut[(size_t)(2 * q + 1)], z = cut[(size_t)(2 * q + 2)];
                        std::merge(cells.begin() + a, cells.begin() + m, cells.begin() + m, cells.begin() + z, tmp.begin() + a,
                                   [](const Cell &x, const Cell &y) { return x.key < y.key; });
                    }
                });
                if ((cut.size() - 1) % 2) // an odd range out: carried over as it is
                    std::copy(cells.begin() + cut[cut.size() - 2], cells.begin() + cut[cut.size() - 1], tmp.begin() + cut[cut.size() - 2]);
                cells.swap(tmp);
                std::vector<int64_t> next;
                for (size_t q = 0; q + 1 < cut.size(); q += 2) next.push_back(cut[q]);
                next.push_back(n_lines);
                cut.swap(next);
            }
            parallel_done = true;
            lap("sort");
        }
    }
    if (!parallel_done) {
    cells.reserve(lines.size());
    std::vector<int32_t> cond_list;
    std::string ctx;
    for (size_t ln = 1; ln < lines.size(); ++ln) {
        const char *ub, *ue, *ib, *ie;
        double rate = 0.0;
        if (!parse_line(ln, ub, ue, ib, ie, rate, ctx, cond_list, d->err)) {
            g_dao_err = d->err;
            delete d;
            return CMI_E_INVALID;
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
        const int32_t cc = d->ctx_ids.find_or_add(ctx.data(), ctx.size(), d->ctxs);
        if ((size_t)cc == d->ctx_cond_list.size()) d->ctx_cond_list.push_back(cond_list);
        else d->ctx_cond_list[(size_t)cc] = cond_list; // contextConditionsList.put(cc, condList): same list by construction
        cells.push_back(Cell{((uint64_t)(uint32_t)uic << 32) | (uint32_t)cc, rate});
    }
    }
    // ratingScale: sorted distinct values (DataDAO.java:348-350)
    std::sort(scale.begin(), scale.end());
    scale.erase(std::unique(scale.begin(), scale.end()), scale.end());
    d->rating_scale = scale;
    // CRS order; a stable sort keeps the file order inside a cell, whose LAST entry wins
    if (!parallel_done) std::stable_sort(cells.begin(), cells.end(), [](const Cell &x, const Cell &y) { return x.key < y.key; });
    // a cell keeps its LAST entry (the next entry holds another key): counted per range, placed per range
    {
        const int64_t nc = (int64_t)cells.size();
        const int ntm = parallel_done ? host_threads(nc) : 1;
        std::vector<int64_t> kept((size_t)ntm + 1, 0);
        auto last_of_cell = [&](int64_t i) { return i + 1 >= nc || cells[(size_t)i + 1].key != cells[(size_t)i].key; };
        parallel_ranges(nc, ntm, [&](int part, int64_t b, int64_t e) {
            int64_t k = 0;
            for (int64_t i = b; i < e; ++i) k += last_of_cell(i);
            kept[(size_t)part + 1] = k;
        });
        for (int t = 0; t < ntm; ++t) kept[(size_t)t + 1] += kept[(size_t)t];
        d->m_ui.resize((size_t)kept[(size_t)ntm]);
        d->m_ctx.resize((size_t)kept[(size_t)ntm]);
        d->m_r.resize((size_t)kept[(size_t)ntm]);
        parallel_ranges(nc, ntm, [&](int part, int64_t b, int64_t e) {
            int64_t k = kept[(size_t)part];
            for (int64_t i = b; i < e; ++i)
                if (last_of_cell(i)) {
                    d->m_ui[(size_t)k] = (int32_t)(cells[(size_t)i].key >> 32);
                    d->m_ctx[(size_t)k] = (int32_t)(cells[(size_t)i].key & 0xffffffffu);
                    d->m_r[(size_t)k] = cells[(size_t)i].rate;
                    ++k;
                }
        });
    }
    lap("matrix");
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

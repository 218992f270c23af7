// rank_api.cpp -- cmi_eval_rankings: the reference's top-N evaluation (Recommender.evalRankings,
// src/carskit/generic/Recommender.java:668-964) with the O(queries x items x k) scoring on the GPU.
// Host side: query / candidate / exclusion bookkeeping and the (tiny, per-query) metric formulas; device side:
// rank_kernels.hip.  No CPU scoring path exists.
#include <algorithm>
#include "env_knobs.hpp"
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <cstring>
#include <future>
#include <memory>
#include <numeric>
#include <thread>

#include <unistd.h>

#include "cmi_instance.hpp"
#include "host_pool.hpp"
#include "rank_host.hpp"
#include "rank_kernels.hpp"

using namespace cmi;

namespace {

// ---- java.util.HashSet<Integer> iteration order -----------------------------------------------------------------
// Integer.hashCode() is the value; HashMap spreads h ^ (h >>> 16), indexes with (cap-1), doubles the table when
// size > 0.75*cap, and on a split keeps relative order inside a bucket, so iteration = (bucket, insertion order).
std::vector<int32_t> java_int_hashset_order(const std::vector<int32_t> &first_seen) {
    size_t cap = 16;
    while ((double)first_seen.size() > 0.75 * (double)cap) cap <<= 1;
    std::vector<std::pair<uint32_t, int32_t>> keyed(first_seen.size());
    for (size_t i = 0; i < first_seen.size(); ++i) {
        const uint32_t h = (uint32_t)first_seen[i];
        keyed[i] = {(uint32_t)((h ^ (h >> 16)) & (cap - 1)), (int32_t)i};
    }
    std::stable_sort(keyed.begin(), keyed.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    std::vector<int32_t> out(first_seen.size());
    for (size_t i = 0; i < keyed.size(); ++i) out[i] = first_seen[keyed[i].second];
    return out;
}

// ---- metric formulas (happy.coding.math.Measures, read from lib/happy.coding.utils-1.2.6.jar; wrappers that cut
// the list to the top n first: src/carskit/eval/Measures.java:13-67) -------------------------------------------------
struct Truth {
    const int32_t *items;
    int n;
    bool has(int32_t j) const { // ground-truth lists are short; sorted for the binary search
        return std::binary_search(items, items + n, j);
    }
};

// The measures below see a list through its membership flags rel[i] = "ranked[i] is a correct item" (one binary search per list
// entry, done once per list: every formula of happy.coding.math.Measures asks only that question about an entry).
int hits_at(const bool *rel, int len, int n) { // Measures.HitsAt over the FULL ranked list
    int hits = 0;
    for (int i = 0; i < len; ++i)
        if (rel[i]) {
            if (i >= n) break;
            ++hits;
        }
    return hits;
}

double auc(const bool *rel, int len, int n_truth, int num_dropped) {
    int num_rele = 0; // Lists.overlapSize(groundTruth, rankedList)
    for (int i = 0; i < len; ++i) num_rele += rel[i];
    const long num_eval_items = (long)len + num_dropped;
    const long num_eval_pairs = (num_eval_items - num_rele) * num_rele;
    if (num_eval_pairs == 0) return 0.5;
    long correct = 0, hits = 0;
    for (int i = 0; i < len; ++i) {
        if (!rel[i]) correct += hits;
        else ++hits;
    }
    const long num_miss = n_truth - num_rele; // Lists.exceptSize(groundTruth, rankedList)
    correct += hits * ((long)num_dropped - num_miss);
    return (double)correct / (double)num_eval_pairs;
}

double ap(const bool *rel, int len, int n_truth) {
    int hits = 0;
    double s = 0.0;
    for (int i = 0; i < len; ++i)
        if (rel[i]) {
            ++hits;
            s += hits / (i + 1.0);
        }
    return hits > 0 ? s / n_truth : 0.0;
}

inline double log2j(double x) { return std::log(x) / std::log(2.0); } // Maths.log(x, 2)
// 1 / log2(i + 2) for the first positions: the same expression evaluated once instead of once per list entry
struct InvLog2 {
    static constexpr int N = 1024;
    double v[N];
    InvLog2() {
        for (int i = 0; i < N; ++i) v[i] = 1.0 / log2j(i + 2);
    }
    double operator()(int i) const { return i < N ? v[i] : 1.0 / log2j(i + 2); }
};
const InvLog2 inv_log2;

double ndcg(const bool *rel, int len, int n_truth) {
    double dcg = 0.0, idcg = 0.0;
    for (int i = 0; i < len; ++i)
        if (rel[i]) dcg += inv_log2(i);
    for (int i = 0; i < n_truth; ++i) idcg += inv_log2(i);
    return dcg / idcg;
}

double rr(const bool *rel, int len) {
    for (int i = 0; i < len; ++i)
        if (rel[i]) return 1.0 / (i + 1.0);
    return 0.0;
}

constexpr int N_MEAS = 18; // Pre,Rec,AUC,MAP,NDCG,MRR x {5,10,N}

// the 18 measures of ONE ranked list (already cut at num_recs): index = measure * 3 + cut-off, cut-offs {5, 10, num_recs}
// (Recommender.java:852-858 through carskit.eval.Measures.*At)
void list_measures(const int32_t *ranked, int len, const Truth &t, int num_dropped, int num_recs, double vals[N_MEAS]) {
    bool rel_small[64];
    std::unique_ptr<bool[]> rel_big;
    bool *rel = rel_small;
    if (len > 64) {
        rel_big.reset(new bool[(size_t)len]);
        rel = rel_big.get();
    }
    for (int i = 0; i < len; ++i) rel[i] = t.has(ranked[i]);
    const int cut[3] = {5, 10, num_recs};
    for (int c = 0; c < 3; ++c) {
        const int n = cut[c], tl = std::min(n, len);
        const int hits = hits_at(rel, len, n);
        vals[0 + c] = hits / (n + 0.0);
        vals[3 + c] = hits / (t.n + 0.0);
        vals[6 + c] = auc(rel, tl, t.n, num_dropped);
        vals[9 + c] = ap(rel, tl, t.n);
        vals[12 + c] = ndcg(rel, tl, t.n);
        vals[15 + c] = rr(rel, tl);
    }
}

} // namespace

namespace cmi {

namespace {
// Host scratch of rank_build_plan, kept between evaluations: the plan moves tens of megabytes through per-thread lists, and a fresh
// allocation pays a page fault per 4 KB touched -- from 16 threads inside one address space those serialise in the kernel and cost more
// than the work itself (measured on an MI355X box: the tuple passes ran at 14 ns per tuple).  One cached instance; a concurrent
// evaluation (another fold on another host thread) builds its own and drops it.
struct PlanUK {
    uint32_t u, c, j;
};
struct PlanPart {
    std::vector<int32_t> qu, qc, truth_items, excl_idx;
    std::vector<int64_t> truth_end, excl_end; // running ends inside this part
    void clear() {
        qu.clear();
        qc.clear();
        truth_items.clear();
        excl_idx.clear();
        truth_end.clear();
        excl_end.clear();
    }
};
template <typename T>
struct RawBuf { // uninitialised, grow-only
    std::unique_ptr<T[]> p;
    size_t cap = 0;
    T *need(size_t n) {
        if (n > cap) {
            p.reset(new T[n + n / 8]);
            cap = n + n / 8;
        }
        return p.get();
    }
};
struct PlanScratch {
    std::vector<std::vector<int32_t>> lfirst, ldeg;
    std::vector<std::vector<PlanUK>> tl, pl;
    std::vector<PlanPart> parts;
    RawBuf<uint64_t> tkey, pkey;
    RawBuf<int64_t> toff, poff;
    size_t bytes() const {
        size_t b = (tkey.cap + pkey.cap + toff.cap + poff.cap) * 8;
        for (const auto &v : tl) b += v.capacity() * sizeof(PlanUK);
        for (const auto &v : pl) b += v.capacity() * sizeof(PlanUK);
        for (const auto &v : ldeg) b += v.capacity() * 4;
        return b;
    }
};
std::mutex plan_scratch_mu;
PlanScratch *plan_scratch_cached = nullptr;
struct PlanScratchLease {
    PlanScratch *s;
    PlanScratchLease() {
        std::lock_guard<std::mutex> g(plan_scratch_mu);
        s = plan_scratch_cached;
        plan_scratch_cached = nullptr;
        if (!s) s = new PlanScratch();
    }
    ~PlanScratchLease() {
        std::lock_guard<std::mutex> g(plan_scratch_mu);
        if (!plan_scratch_cached && s->bytes() <= ((size_t)1 << 30)) plan_scratch_cached = s;
        else delete s;
    }
};
} // namespace

void rank_build_plan(int n_users, int n_items, const RankTuples &train, const RankTuples &test, double bin_thold,
                     int num_ignore, RankPlan &plan) {
    const int nt = host_threads(train.n + test.n);
    const bool times = getenv("CMI_PLAN_TIMES") != nullptr; // tools/exp/rank_plan_time.py
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *w) {
        if (!times) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "plan %s %.3f ms\n", w, std::chrono::duration<double, std::milli>(n - T0).count());
        T0 = n;
    };
    auto range_of = [](int64_t n, int parts, int p, int64_t &b, int64_t &e) {
        const int64_t step = (n + parts - 1) / parts;
        b = std::min<int64_t>(n, p * step);
        e = std::min<int64_t>(n, b + step);
    };
    PlanScratchLease lease;
    PlanScratch &S = *lease.s;
    const size_t nu = (size_t)n_users;
    // users fall into at most 256 BUCKETS of 2^sh consecutive ids (a shift per tuple, not a division)
    int sh = 0;
    // (256 buckets measured best on an MI355X box's host: 3.7 ms for the 2 M tuples of the bench against 4.1 / 5.1 with 64 / 16)
    while ((((int64_t)n_users - 1) >> sh) >= 256) ++sh;
    const int nbk = (int)(((int64_t)n_users - 1) >> sh) + 1;
    using UK = PlanUK;
    std::vector<std::vector<UK>> &tl = S.tl, &pl = S.pl; // [tuple range][user bucket]
    if (tl.size() < (size_t)nt * nbk) tl.resize((size_t)nt * nbk);
    if (pl.size() < (size_t)nt * nbk) pl.resize((size_t)nt * nbk);
    int np = (int)std::max<int64_t>(1, std::min<int64_t>(nt, ((int64_t)16 << 20) / std::max(n_items, 1)));
    if (const char *e = getenv("CMI_PLAN_DEG_RANGES")) np = std::max(1, std::min(atoi(e), nt)); // tests: the huge-catalogue form on small inputs
    std::vector<std::vector<int32_t>> &lfirst = S.lfirst, &ldeg = S.ldeg;
    if (lfirst.size() < (size_t)nt) lfirst.resize((size_t)nt);
    if (ldeg.size() < (size_t)nt) ldeg.resize((size_t)nt);
    auto positive = [&](int64_t t) { return test.r[t] != 0.0 && test.r[t] > bin_thold; };

    // ONE pass over the tuples, in ranges of tuples.  Per range: (a) the items in first-seen order with their degrees (candidates:
    // rateDao.getItemList(trainMatrix) -> HashSet<Integer>, DataDAO.java:1210-1218, whose iteration order is computed from the
    // first-seen order); (b) the training tuples and the test positives (rate > threshold; DataDAO.getUserCtxList,
    // DataDAO.java:1114-1140) appended to one sequential list per user bucket.
    parallel_ranges(nt, nt, [&](int, int64_t p0, int64_t p1) {
        for (int64_t p = p0; p < p1; ++p) {
            std::vector<UK> *tp = &tl[(size_t)p * nbk], *pp = &pl[(size_t)p * nbk];
            std::vector<int32_t> &deg = ldeg[(size_t)p], &fs = lfirst[(size_t)p];
            fs.clear();
            int64_t b, e;
            range_of(train.n, nt, (int)p, b, e);
            for (int q = 0; q < nbk; ++q) {
                tp[q].clear();
                pp[q].clear();
                tp[q].reserve((size_t)((e - b) / nbk + (e - b) / (4 * nbk) + 16));
            }
            const bool fused = np == nt; // every tuple range records its own degrees: no second pass over the items
            if (fused) deg.assign((size_t)n_items, 0);
            for (int64_t t = b; t < e; ++t) {
                if (train.r && train.r[t] == 0.0) continue; // a sparse matrix holds no zero entries
                const uint32_t u = (uint32_t)train.u[t];
                const int32_t j = train.j[t];
                tp[u >> sh].push_back(UK{u, (uint32_t)train.ctx[t], (uint32_t)j});
                if (fused && deg[j]++ == 0) fs.push_back(j);
            }
            range_of(test.n, nt, (int)p, b, e);
            for (int64_t t = b; t < e; ++t)
                if (positive(t)) {
                    const uint32_t u = (uint32_t)test.u[t];
                    pp[u >> sh].push_back(UK{u, (uint32_t)test.ctx[t], (uint32_t)test.j[t]});
                }
        }
    });
    if (np != nt) // a huge catalogue: per-range degrees cost n_items ints each, so fewer (larger) ranges record them
        parallel_ranges(np, np, [&](int, int64_t p0, int64_t p1) {
            for (int64_t p = p0; p < p1; ++p) {
                std::vector<int32_t> &deg = ldeg[(size_t)p], &fs = lfirst[(size_t)p];
                deg.assign((size_t)n_items, 0);
                int64_t b, e;
                range_of(train.n, np, (int)p, b, e);
                for (int64_t t = b; t < e; ++t) {
                    if (train.r && train.r[t] == 0.0) continue;
                    if (deg[train.j[t]]++ == 0) fs.push_back(train.j[t]);
                }
            }
        });
    lap("tuples");
    // An item's global first occurrence lies in the EARLIEST range that holds it: per item the earliest range and the total degree
    // (ranges of items), then every range keeps the items it saw first (in its own order), and the ranges follow each other.
    std::vector<int32_t> first_seen, degree(n_items, 0), pmin(n_items, -1);
    parallel_ranges(n_items, np, [&](int, int64_t j0, int64_t j1) {
        for (int p = 0; p < np; ++p) {
            const int32_t *d = ldeg[(size_t)p].data();
            for (int64_t j = j0; j < j1; ++j)
                if (d[j]) {
                    if (pmin[(size_t)j] < 0) pmin[(size_t)j] = p;
                    degree[(size_t)j] += d[j];
                }
        }
    });
    for (int p = 0; p < np; ++p)
        for (int32_t j : lfirst[(size_t)p])
            if (pmin[(size_t)j] == p) first_seen.push_back(j);
    lap("first_seen");
    std::vector<int32_t> &cand = plan.cand;
    cand = java_int_hashset_order(first_seen);
    if (num_ignore > 0) { // drop the most popular items (Recommender.java:720-735): stable sort by degree, descending
        std::vector<int32_t> by_deg = cand;
        std::stable_sort(by_deg.begin(), by_deg.end(), [&](int32_t a, int32_t b) { return degree[a] > degree[b]; });
        std::vector<char> drop(n_items, 0);
        for (int i = 0; i < num_ignore && i < (int)by_deg.size(); ++i) drop[by_deg[i]] = 1;
        cand.erase(std::remove_if(cand.begin(), cand.end(), [&](int32_t j) { return drop[j]; }), cand.end());
    }
    const int nc = (int)cand.size();
    std::vector<int32_t> cand_pos(n_items, -1);
    for (int i = 0; i < nc; ++i) cand_pos[cand[i]] = i;
    lap("cand");

    // Every user BUCKET walks its lists in tuple-range order (that keeps the caller's order inside a user), counts, and places the
    // packed (context, item) keys inside its own region of the key array (a region and its cursors fit the core's cache; no two
    // threads share a cursor).
    std::vector<int64_t> tbase((size_t)nbk + 1, 0), pbase((size_t)nbk + 1, 0); // first key of every bucket
    for (int q = 0; q < nbk; ++q) {
        int64_t ts = 0, ps = 0;
        for (int p = 0; p < nt; ++p) {
            ts += (int64_t)tl[(size_t)p * nbk + q].size();
            ps += (int64_t)pl[(size_t)p * nbk + q].size();
        }
        tbase[(size_t)q + 1] = tbase[(size_t)q] + ts;
        pbase[(size_t)q + 1] = pbase[(size_t)q] + ps;
    }
    int64_t *toff = S.toff.need(nu + 1), *poff = S.poff.need(nu + 1);
    toff[nu] = tbase[(size_t)nbk];
    poff[nu] = pbase[(size_t)nbk];
    uint64_t *tkey = S.tkey.need((size_t)toff[nu] + 1), *pkey = S.pkey.need((size_t)poff[nu] + 1);
    // per user: its queries in (context, item) order -- a query is a (user, context) with at least one correct item that is a
    // candidate (Recommender.java:789-790) -- and, per query, the candidate positions of the items the user rated in the same
    // context in the training set (Recommender.java:793, 814-816), as ascending candidate positions.  One part per bucket.
    using Part = PlanPart;
    std::vector<Part> &parts = S.parts;
    if (parts.size() < (size_t)nbk) parts.resize((size_t)nbk);
    parallel_ranges(nbk, nt, [&](int, int64_t k0, int64_t k1) {
        std::vector<int64_t> cur;
        std::vector<uint32_t> items;
        for (int64_t k = k0; k < k1; ++k) {
            const int64_t u0 = k << sh, u1 = std::min<int64_t>(n_users, (k + 1) << sh);
            auto place = [&](const std::vector<std::vector<UK>> &lists, int64_t base, int64_t *off, uint64_t *key) {
                cur.assign((size_t)(u1 - u0) + 1, 0);
                for (int p = 0; p < nt; ++p)
                    for (const UK &x : lists[(size_t)p * nbk + (size_t)k]) cur[(size_t)(x.u - u0) + 1]++;
                int64_t run = base;
                for (int64_t u = u0; u < u1; ++u) {
                    off[(size_t)u] = run;
                    const int64_t n = cur[(size_t)(u - u0) + 1];
                    cur[(size_t)(u - u0)] = run;
                    run += n;
                }
                for (int p = 0; p < nt; ++p)
                    for (const UK &x : lists[(size_t)p * nbk + (size_t)k]) key[(size_t)cur[(size_t)(x.u - u0)]++] = ((uint64_t)x.c << 32) | x.j;
            };
            place(tl, tbase[(size_t)k], toff, tkey);
            place(pl, pbase[(size_t)k], poff, pkey);
            Part &P = parts[(size_t)k];
            P.clear();
            for (int64_t u = u0; u < u1; ++u) {
                uint64_t *pb = pkey + poff[(size_t)u], *pe = u + 1 < u1 ? pkey + poff[(size_t)u + 1] : pkey + pbase[(size_t)k + 1];
                if (pb == pe) continue;
                std::sort(pb, pe);
                const uint64_t *tb = tkey + toff[(size_t)u], *te = u + 1 < u1 ? tkey + toff[(size_t)u + 1] : tkey + tbase[(size_t)k + 1];
                for (uint64_t *i = pb; i < pe;) {
                    const uint32_t c = (uint32_t)(*i >> 32);
                    const size_t before = P.truth_items.size();
                    uint64_t *e = i;
                    for (; e < pe && (uint32_t)(*e >> 32) == c; ++e) {
                        const int32_t j = (int32_t)(uint32_t)*e;
                        if (cand_pos[j] >= 0 && (P.truth_items.size() == before || P.truth_items.back() != j)) P.truth_items.push_back(j);
                    }
                    i = e;
                    if (P.truth_items.size() == before) continue;
                    P.qu.push_back((int32_t)u);
                    P.qc.push_back((int32_t)c);
                    P.truth_end.push_back((int64_t)P.truth_items.size());
                    const size_t ebefore = P.excl_idx.size();
                    items.clear(); // the user's training items in this context (users hold tens of tuples: a scan per query) ...
                    for (const uint64_t *t = tb; t < te; ++t)
                        if ((uint32_t)(*t >> 32) == c) items.push_back((uint32_t)*t);
                    // ... as candidate POSITIONS in ascending order, each once: the selection walks the list beside the candidates
                    // (with sparse item ids the candidates' HashSet order is not the order of the ids)
                    for (uint32_t j : items) {
                        const int32_t cp = cand_pos[j];
                        if (cp >= 0) P.excl_idx.push_back(cp);
                    }
                    std::sort(P.excl_idx.begin() + (std::ptrdiff_t)ebefore, P.excl_idx.end());
                    P.excl_idx.erase(std::unique(P.excl_idx.begin() + (std::ptrdiff_t)ebefore, P.excl_idx.end()), P.excl_idx.end());
                    P.excl_end.push_back((int64_t)P.excl_idx.size());
                }
            }
        }
    });
    lap("users");
    // the parts, in bucket order
    std::vector<int64_t> q0((size_t)nbk + 1, 0), t0((size_t)nbk + 1, 0), e0((size_t)nbk + 1, 0), g0((size_t)nbk + 1, 0);
    parallel_ranges(nbk, nt, [&](int, int64_t p0, int64_t p1) { // query groups (runs of one user) per part: users never straddle parts
        for (int64_t p = p0; p < p1; ++p) {
            const std::vector<int32_t> &qu = parts[(size_t)p].qu;
            int64_t g = 0;
            for (size_t i = 0; i < qu.size(); ++i) g += i == 0 || qu[i] != qu[i - 1];
            g0[(size_t)p + 1] = g;
        }
    });
    for (int p = 0; p < nbk; ++p) {
        g0[(size_t)p + 1] += g0[(size_t)p];
        q0[(size_t)p + 1] = q0[(size_t)p] + (int64_t)parts[(size_t)p].qu.size();
        t0[(size_t)p + 1] = t0[(size_t)p] + (int64_t)parts[(size_t)p].truth_items.size();
        e0[(size_t)p + 1] = e0[(size_t)p] + (int64_t)parts[(size_t)p].excl_idx.size();
    }
    plan.qu.resize((size_t)q0[(size_t)nbk]);
    plan.qc.resize((size_t)q0[(size_t)nbk]);
    plan.truth_ptr.resize((size_t)q0[(size_t)nbk] + 1);
    plan.excl_ptr.resize((size_t)q0[(size_t)nbk] + 1);
    plan.truth_items.resize((size_t)t0[(size_t)nbk]);
    plan.excl_idx.resize((size_t)e0[(size_t)nbk]);
    plan.truth_ptr[0] = plan.excl_ptr[0] = 0;
    plan.qg.resize((size_t)q0[(size_t)nbk]);
    plan.gu.resize((size_t)g0[(size_t)nbk]);
    plan.gq0.resize((size_t)g0[(size_t)nbk] + 1);
    plan.gq0[(size_t)g0[(size_t)nbk]] = (int32_t)q0[(size_t)nbk];
    parallel_ranges(nbk, nt, [&](int, int64_t p0, int64_t p1) {
        for (int64_t p = p0; p < p1; ++p) {
            const Part &P = parts[(size_t)p];
            int64_t g = g0[(size_t)p] - 1;
            for (size_t i = 0; i < P.qu.size(); ++i) {
                if (i == 0 || P.qu[i] != P.qu[i - 1]) {
                    ++g;
                    plan.gu[(size_t)g] = P.qu[i];
                    plan.gq0[(size_t)g] = (int32_t)(q0[(size_t)p] + (int64_t)i);
                }
                plan.qg[(size_t)q0[(size_t)p] + i] = (int32_t)g;
            }
            std::copy(P.qu.begin(), P.qu.end(), plan.qu.begin() + q0[(size_t)p]);
            std::copy(P.qc.begin(), P.qc.end(), plan.qc.begin() + q0[(size_t)p]);
            std::copy(P.truth_items.begin(), P.truth_items.end(), plan.truth_items.begin() + t0[(size_t)p]);
            std::copy(P.excl_idx.begin(), P.excl_idx.end(), plan.excl_idx.begin() + e0[(size_t)p]);
            for (size_t i = 0; i < P.truth_end.size(); ++i) {
                plan.truth_ptr[(size_t)q0[(size_t)p] + i + 1] = t0[(size_t)p] + P.truth_end[i];
                plan.excl_ptr[(size_t)q0[(size_t)p] + i + 1] = e0[(size_t)p] + P.excl_end[i];
            }
        }
    });
    lap("concat");
}

void rank_measures_range(const RankPlan &plan, int num_recs, const int32_t *top_idx, const double *top_score, const int32_t *top_count,
                         int64_t q_begin, int64_t q_end, double *vals, int32_t *q_user, int32_t *q_ctx, int32_t *q_count,
                         int32_t *top_items, double *top_scores) {
    const int nc = (int)plan.cand.size();
    // each query's measures are a pure function of its list: ranges of queries on the host's cores
    parallel_ranges(q_end - q_begin, host_threads((q_end - q_begin) * 8), [&](int, int64_t r0, int64_t r1) {
        std::vector<int32_t> ranked(num_recs);
        for (int64_t q = q_begin + r0; q < q_begin + r1; ++q) {
            const int len = top_count[q];
            if (q_user) q_user[q] = plan.qu[q];
            if (q_ctx) q_ctx[q] = plan.qc[q];
            if (q_count) q_count[q] = len;
            for (int i = 0; i < len; ++i) {
                ranked[i] = plan.cand[top_idx[(size_t)q * num_recs + i]];
                if (top_items) top_items[(size_t)q * num_recs + i] = ranked[i];
                if (top_scores) top_scores[(size_t)q * num_recs + i] = top_score[(size_t)q * num_recs + i];
            }
            for (int i = len; i < num_recs; ++i) {
                if (top_items) top_items[(size_t)q * num_recs + i] = -1;
                if (top_scores) top_scores[(size_t)q * num_recs + i] = std::nan("");
            }
            if (len > 0) { // "no recommendations available" queries are skipped (Recommender.java:818-819)
                const Truth t{plan.truth_items.data() + plan.truth_ptr[q], (int)(plan.truth_ptr[q + 1] - plan.truth_ptr[q])};
                const int num_cands = nc - (int)(plan.excl_ptr[q + 1] - plan.excl_ptr[q]);
                list_measures(ranked.data(), len, t, num_cands - len, num_recs, vals + (size_t)q * N_MEAS);
            }
        }
    });
}

namespace {
// Every addition of the averages is a happy.coding.math.Stats.mean step (NaN entries are skipped; an empty mean is 0/0 = NaN), in query
// order.  The 18 chains are independent of each other, so they advance side by side without branches: a skipped entry adds +0.0 (the
// sums are never -0.0: they start at +0.0 and +0.0 + -0.0 = +0.0) and counts 0.
struct MeanAcc {
    double s[N_MEAS];
    int64_t c[N_MEAS];
    MeanAcc() {
        for (int m = 0; m < N_MEAS; ++m) {
            s[m] = 0.0;
            c[m] = 0;
        }
    }
    void add(const double *v) {
        for (int m = 0; m < N_MEAS; ++m) {
            const bool ok = v[m] == v[m];
            s[m] += ok ? v[m] : 0.0;
            c[m] += ok;
        }
    }
    void mean(double *o) const {
        for (int m = 0; m < N_MEAS; ++m) o[m] = c[m] ? s[m] / (double)c[m] : std::nan("");
    }
};
static_assert(N_MEAS == 18, "RankFolded::s / c (rank_host.hpp) hold one chain per measure");
// the running sums of a RankFolded advance exactly as one MeanAcc over all rows would: same rows, same order, same operations
inline void folded_add(RankFolded &f, const double *v) {
    for (int m = 0; m < N_MEAS; ++m) {
        const bool ok = v[m] == v[m];
        f.s[m] += ok ? v[m] : 0.0;
        f.c[m] += ok;
    }
}
} // namespace

void rank_sum_queries(const int32_t *top_count, const double *vals, RankFolded &f, int64_t q_to) {
    for (int64_t q = f.summed; q < q_to; ++q)
        if (top_count[q] > 0) folded_add(f, vals + (size_t)q * N_MEAS);
    f.summed = std::max(f.summed, q_to);
}

// ucu: a test user contributes the NaN-skipping mean over its contexts -- NaN (skipped again) if none of its contexts produced a list
// (Recommender.java:903-926).  The users' means do not depend on each other: ranges of whole users on the host's cores compute them and
// write them, in user order, into `umeans` (18 doubles per user); only the sum over the users is serial (rank_average).  Users whose
// queries all lie in [f.q, q_to) are folded (f.q is a user's first query); f advances to the first user that was not -- q_to itself when
// `last`.
void rank_fold_users(const RankPlan &plan, const int32_t *top_count, const double *vals, double *umeans, RankFolded &f, int64_t q_to, bool last) {
    const int64_t nq = (int64_t)plan.qu.size(), q_from = f.q;
    int64_t stop = std::min(q_to, nq);
    if (!last && stop < nq)
        while (stop > q_from && plan.qu[(size_t)stop] == plan.qu[(size_t)stop - 1]) --stop; // the user that straddles q_to waits
    if (stop <= q_from) return;
    const int64_t n = stop - q_from;
    const int nt = host_threads(n * 4);
    std::vector<int64_t> cut((size_t)nt + 1, stop), ubase((size_t)nt + 1, 0);
    cut[0] = q_from;
    for (int t = 1; t < nt; ++t) {
        int64_t b = std::min<int64_t>(stop, q_from + (n + nt - 1) / nt * t);
        while (b > q_from && b < stop && plan.qu[(size_t)b] == plan.qu[(size_t)b - 1]) ++b; // to the next user's first query
        cut[(size_t)t] = std::max(b, cut[(size_t)t - 1]);
    }
    parallel_ranges(nt, nt, [&](int, int64_t p0, int64_t p1) { // users per range
        for (int64_t p = p0; p < p1; ++p) {
            int64_t c = 0;
            for (int64_t q = cut[(size_t)p]; q < cut[(size_t)p + 1]; ++q) c += q + 1 == nq || plan.qu[(size_t)q + 1] != plan.qu[(size_t)q];
            ubase[(size_t)p + 1] = c;
        }
    });
    ubase[0] = f.u;
    for (int t = 0; t < nt; ++t) ubase[(size_t)t + 1] += ubase[(size_t)t];
    parallel_ranges(nt, nt, [&](int, int64_t p0, int64_t p1) {
        for (int64_t p = p0; p < p1; ++p) {
            MeanAcc user;
            double *dst = umeans + (size_t)ubase[(size_t)p] * N_MEAS;
            for (int64_t q = cut[(size_t)p]; q < cut[(size_t)p + 1]; ++q) {
                if (top_count[q] > 0) user.add(vals + (size_t)q * N_MEAS);
                if (q + 1 == nq || plan.qu[(size_t)q + 1] != plan.qu[(size_t)q]) {
                    user.mean(dst);
                    dst += N_MEAS;
                    user = MeanAcc();
                }
            }
        }
    });
    f.q = stop;
    f.u = ubase[(size_t)nt];
    // the sum over the users is serial (its order is the result's last bits): taken here, batch by batch, for the users just folded
    for (int64_t u = f.summed; u < f.u; ++u) folded_add(f, umeans + (size_t)u * N_MEAS);
    f.summed = f.u;
}

void rank_average(const RankPlan &plan, int strategy, const int32_t *top_count, const double *vals, double *umeans, RankFolded f, double *out) {
    const int64_t nq = (int64_t)plan.qu.size();
    for (int m = 0; m < CMI_RANK_MEASURES; ++m) out[m] = std::nan("");
    out[18] = out[19] = out[20] = 0.0; // D5/D10/DN: isDiverseUsed=false (Recommender.java:939-941)
    if (strategy == CMI_RANK_UC) {
        rank_sum_queries(top_count, vals, f, nq); // (what the batches have not added yet)
    } else {
        std::unique_ptr<double[]> own;
        if (!umeans) { // a caller without a workspace buffer: at most one user per query
            own.reset(new double[(size_t)nq * N_MEAS + 1]);
            umeans = own.get();
            f = RankFolded();
        }
        if (f.q < nq) rank_fold_users(plan, top_count, vals, umeans, f, nq, true); // (also sums the users it folds)
    }
    MeanAcc total;
    for (int m = 0; m < N_MEAS; ++m) total.s[m] = f.s[m], total.c[m] = f.c[m];
    total.mean(out);
}

hipError_t RankWorkspace::need(Buf &b, size_t bytes, bool pinned) {
    bytes = std::max<size_t>(bytes, 8);
    if (b.cap >= bytes) return hipSuccess;
    if (b.p) (void)(pinned ? hipHostFree(b.p) : hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    const size_t want = bytes + bytes / 8; // a little head room: the next fold's test set is rarely the same size
    hipError_t e = pinned ? hipHostMalloc(&b.p, want, hipHostMallocDefault) : hipMalloc(&b.p, want);
    if (e == hipSuccess) b.cap = want;
    return e;
}

void RankWorkspace::release() {
    Buf *dev[] = {&dA, &dB, &dS, &drc, &dcand, &dqu, &dqc, &dexptr, &dexcl, &dtop, &dscore, &dcount, &dB2, &dA2, &dS2, &dqg, &dqd, &dgu, &ddc, &dscr, &dSb, &dAb, &dcolc, &dM1, &dM1b, &dM2};
    if (sel_stream) (void)hipStreamDestroy(sel_stream);
    sel_stream = nullptr;
    if (gemm_stream) (void)hipStreamDestroy(gemm_stream);
    gemm_stream = nullptr;
    if (ev_gs) (void)hipEventDestroy(ev_gs);
    ev_gs = nullptr;
    for (std::vector<hipEvent_t> *v : {&evgemm, &evsel}) {
        for (hipEvent_t ev : *v) (void)hipEventDestroy(ev);
        v->clear();
    }
    plan_valid = false;
    for (const void *&r : resident) r = nullptr;
    for (Buf *b : dev) {
        if (b->p) (void)hipFree(b->p);
        *b = Buf();
    }
    Buf *host[] = {&h_top, &h_score, &h_count};
    for (Buf *b : host) {
        if (b->p) (void)hipHostFree(b->p);
        *b = Buf();
    }
    hipEvent_t *evs[] = {&ev0, &ev1};
    for (hipEvent_t *e : evs) {
        if (*e) (void)hipEventDestroy(*e);
        *e = nullptr;
    }
    for (hipEvent_t e : evb)
        if (e) (void)hipEventDestroy(e);
    evb.clear();
    for (hipEvent_t e : evk)
        if (e) (void)hipEventDestroy(e);
    evk.clear();
}

hipError_t RankWorkspace::kernel_event(size_t i, hipStream_t stream) {
    while (evk.size() <= i) {
        hipEvent_t ev = nullptr;
        if (hipError_t e = hipEventCreate(&ev)) return e;
        evk.push_back(ev);
    }
    return hipEventRecord(evk[i], stream);
}

hipError_t RankWorkspace::batch_event(size_t b, hipStream_t stream) {
    while (evb.size() <= b) {
        hipEvent_t ev = nullptr;
        if (hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) return e;
        evb.push_back(ev);
    }
    return hipEventRecord(evb[b], stream);
}

hipError_t RankWorkspace::consume_batches(hipError_t e, const std::vector<std::pair<int64_t, int64_t>> &batches,
                                          const std::function<void(int64_t, int64_t)> &on_batch,
                                          std::chrono::steady_clock::time_point t_loop) {
    const bool times = getenv("CMI_PLAN_TIMES") != nullptr;
    auto at = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop).count(); };
    if (times) fprintf(stderr, "rank loop: all batches enqueued at %.3f ms\n", at());
    for (size_t b = 0; b + 1 < batches.size() && e == hipSuccess; ++b) {
        e = hipEventSynchronize(evb[b]);
        if (times) fprintf(stderr, "rank loop: batch %zu's lists on the host at %.3f ms\n", b, at());
        if (e == hipSuccess && on_batch) on_batch(batches[b].first, batches[b].second);
    }
    if (times) fprintf(stderr, "rank loop: host done with all but the last batch at %.3f ms\n", at());
    if (e == hipSuccess) e = hipEventSynchronize(ev1); // recorded behind the last batch's copies
    if (times) fprintf(stderr, "rank loop: last batch's lists on the host at %.3f ms\n", at());
    host_ms[2] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop).count();
    const auto t_tail = std::chrono::steady_clock::now();
    if (e == hipSuccess && !batches.empty() && on_batch) on_batch(batches.back().first, batches.back().second);
    host_ms[3] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_tail).count();
    return e;
}

static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// Batch boundaries over n units of at most b.  The host turns a batch's lists into measures while the device works on the batches
// behind it, so the END of the sequence tapers: b / 3, b / 6, b / 12 -- every batch's host work (about half its device time on an
// MI355X box) is covered by the device time of the batches that follow, and only the last, smallest batch's is exposed.  (Until round
// 6 only the final batch was short, b / 16: the batch before it finished on the host 0.3-0.5 ms after the device had gone idle.)
// In front of the taper: equal batches.
static std::vector<int64_t> batch_cuts(int64_t n, int64_t b) {
    if (n <= 0) return {0}; // (no batches)
    std::vector<int64_t> tail;
    int64_t left = n;
    if (b / 12 >= 64)
        for (int64_t s = b / 12; tail.size() < 3 && left > 2 * s; s *= 2) {
            tail.push_back(s);
            left -= s;
        }
    const int64_t m = std::max<int64_t>(1, (left + b - 1) / b);
    std::vector<int64_t> cuts;
    for (int64_t i = 0; i < m; ++i) cuts.push_back(left * i / m); // (pieces of at most ceil(left / m) <= b)
    int64_t x = left;
    for (size_t i = tail.size(); i-- > 0;) {
        cuts.push_back(x);
        x += tail[i];
    }
    cuts.push_back(n);
    return cuts;
}

template <typename T>
hipError_t rank_run_device(hipStream_t stream, RankWorkspace &ws, const RankPlan &plan, const RankOperands<T> &ops, double thold, int topn,
                           const std::function<void(int64_t, int64_t)> &on_batch, float *ms, double *flops) {
    const auto t_setup = std::chrono::steady_clock::now();
    const std::vector<int32_t> &cand = plan.cand, &qu = plan.qu, &qc = plan.qc, &excl_idx = plan.excl_idx;
    const std::vector<int64_t> &excl_ptr = plan.excl_ptr;
    const int nc = (int)cand.size();
    const int64_t nq = (int64_t)qu.size();
    // operand rows are zero-padded to the GEMM's k step; row counts rounded up to the 128-row block tile (pad rows are
    // never written back)
    const int kp = (ops.k_logical + 15) / 16 * 16; // multiple of both kernels' k step
    auto up128 = [](int64_t v) { return (size_t)((v + 127) / 128 * 128); };
    // query batch: keep the score slab around 1 GiB
    int64_t bq = std::max<int64_t>(64, ((int64_t)1 << 30) / ((int64_t)nc * (int64_t)sizeof(T)));
    bq = std::min<int64_t>(bq, nq);
    if (const char *e = getenv("CMI_RANK_BATCH")) bq = std::max<int64_t>(1, std::min<int64_t>(atoll(e), nq)); // tests: batching is invisible

    hipError_t e = hipSuccess;
    auto need = [&](RankWorkspace::Buf &b, size_t bytes, bool pinned = false) {
        if (e == hipSuccess) e = ws.need(b, bytes, pinned);
    };
    need(ws.dB, up128(nc) * kp * sizeof(T));
    need(ws.dA, up128(bq) * kp * sizeof(T));
    need(ws.dS, (size_t)bq * (size_t)nc * sizeof(T));
    need(ws.drc, (size_t)bq * sizeof(T));
    need(ws.dcand, (size_t)nc * 4);
    need(ws.dqu, (size_t)nq * 4);
    need(ws.dqc, (size_t)nq * 4);
    need(ws.dexptr, (size_t)(nq + 1) * 8);
    need(ws.dexcl, excl_idx.size() * 4);
    need(ws.dtop, (size_t)nq * topn * 4);
    need(ws.dscore, (size_t)nq * topn * 8);
    need(ws.dcount, (size_t)nq * 4);
    need(ws.h_top, (size_t)nq * topn * 4, true);
    need(ws.h_score, (size_t)nq * topn * 8, true);
    need(ws.h_count, (size_t)nq * 4, true);
    hipEvent_t *evs[] = {&ws.ev0, &ws.ev1};
    for (hipEvent_t *ev : evs)
        if (e == hipSuccess && !*ev) e = hipEventCreate(ev);
    if (e != hipSuccess) return e;
    T *dA = (T *)ws.dA.p, *dB = (T *)ws.dB.p, *dS = (T *)ws.dS.p, *drc = (T *)ws.drc.p;
    int32_t *dcand = (int32_t *)ws.dcand.p, *dqu = (int32_t *)ws.dqu.p, *dqc = (int32_t *)ws.dqc.p, *dexcl = (int32_t *)ws.dexcl.p;
    int32_t *dtop = (int32_t *)ws.dtop.p, *dcount = (int32_t *)ws.dcount.p;
    int64_t *dexptr = (int64_t *)ws.dexptr.p;
    double *dscore = (double *)ws.dscore.p;
    auto up = [&](void *d, const void *s, size_t bytes) {
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, stream);
    };
    up(dcand, cand.data(), (size_t)nc * 4);
    up(dqu, qu.data(), (size_t)nq * 4);
    up(dqc, qc.data(), (size_t)nq * 4);
    up(dexptr, excl_ptr.data(), (size_t)(nq + 1) * 8);
    up(dexcl, excl_idx.data(), excl_idx.size() * 4);
    if (e == hipSuccess) e = hipMemsetAsync(dtop, 0xff, (size_t)nq * topn * 4, stream);
    if (e == hipSuccess) e = hipMemsetAsync(dscore, 0, (size_t)nq * topn * 8, stream);
    if (e == hipSuccess) e = ops.build_items(dB, dcand, nc, kp, stream);
    if (e == hipSuccess) e = hipEventRecord(ws.ev0, stream);
    ws.host_ms[1] = ms_since(t_setup);
    const auto t_loop = std::chrono::steady_clock::now();
    // Every batch is enqueued before the host waits for anything (the stream orders a batch's contraction behind the previous
    // batch's selection, which reads the same slab); the host then turns the lists into measures batch by batch as their copies land,
    // behind the device, and can never hold the device up.
    std::vector<std::pair<int64_t, int64_t>> batches;
    const std::vector<int64_t> cuts = batch_cuts(nq, bq);
    for (size_t b = 0; b + 1 < cuts.size() && e == hipSuccess; ++b) {
        const int64_t q0 = cuts[b];
        const int n = (int)(cuts[b + 1] - q0);
        e = ops.build_queries(dA, drc, dqu + q0, dqc + q0, n, kp, stream);
        if (e == hipSuccess) e = rank_launch_score<T>(dA, dB, drc, dS, n, nc, kp, dexptr, dexcl, (int)q0, thold, topn, dtop, dscore, dcount, stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync((int32_t *)ws.h_top.p + (size_t)q0 * topn, dtop + (size_t)q0 * topn, (size_t)n * topn * 4, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync((double *)ws.h_score.p + (size_t)q0 * topn, dscore + (size_t)q0 * topn, (size_t)n * topn * 8, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipMemcpyAsync((int32_t *)ws.h_count.p + q0, dcount + q0, (size_t)n * 4, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = ws.batch_event(batches.size(), stream);
        batches.emplace_back(q0, q0 + n);
    }
    if (e == hipSuccess) e = hipEventRecord(ws.ev1, stream);
    e = ws.consume_batches(e, batches, on_batch, t_loop);
    if (e == hipSuccess && ms) e = hipEventElapsedTime(ms, ws.ev0, ws.ev1);
    if (flops) *flops = 2.0 * (double)nq * (double)nc * (double)kp;
    return e;
}

// ---- the split form (MF family, fp32 state): see rank_kernels.hip "the split form" ---------------------------------------------------------
namespace {
// distinct contexts of the queries (ascending context id) and the dense index of every query's context: O(queries + contexts)
void distinct_contexts(const std::vector<int32_t> &qc, std::vector<int32_t> &dctx, std::vector<int32_t> &qd) {
    int32_t mx = -1;
    for (int32_t c : qc) mx = std::max(mx, c);
    std::vector<int32_t> idx((size_t)mx + 2, -1);
    for (int32_t c : qc) idx[(size_t)c] = 0;
    dctx.clear();
    for (int32_t c = 0; c <= mx; ++c)
        if (idx[(size_t)c] == 0) {
            idx[(size_t)c] = (int32_t)dctx.size();
            dctx.push_back(c);
        }
    qd.resize(qc.size());
    for (size_t i = 0; i < qc.size(); ++i) qd[i] = idx[(size_t)qc[i]];
}
} // namespace

bool rank_split_usable(const RankPlan &plan, int topn, RankWorkspace &ws) {
    ws.ctx_ready = false;
    if (topn > 64 || plan.qu.empty() || plan.cand.empty() || getenv("CMI_RANK_NO_SPLIT")) return false;
    // S2 = distinct contexts x candidates x 4 bytes must stay small (it is re-read by every query of the context); the index arrays
    // computed for the answer are the ones rank_run_device_split uploads
    distinct_contexts(plan.qc, ws.v_dctx, ws.v_qd);
    ws.ctx_ready = true;
    return (int64_t)ws.v_dctx.size() * (int64_t)plan.cand.size() * 4 <= ((int64_t)512 << 20);
}

hipError_t rank_run_device_split(hipStream_t stream, RankWorkspace &ws, const RankPlan &plan, RankSplitArgs a, double thold, int topn,
                                 const std::function<void(int64_t, int64_t)> &on_batch, float *ms, double *flops) {
    const auto t_setup = std::chrono::steady_clock::now();
    const bool times = getenv("CMI_PLAN_TIMES") != nullptr;
    auto tl = t_setup;
    auto lap = [&](const char *w) {
        if (!times) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "rank setup %s %.3f ms\n", w, std::chrono::duration<double, std::milli>(n - tl).count());
        tl = n;
    };
    const int nc = (int)plan.cand.size();
    const int64_t nq = (int64_t)plan.qu.size();
    // query groups = runs of one user (the plan orders the queries by user, then context)
    // (built by the plan's last phase, in ranges)
    const std::vector<int32_t> &gu = plan.gu, &qg = plan.qg, &gq0 = plan.gq0; // group -> user; query -> group; group -> first query
    const int64_t ng = (int64_t)gu.size();
    lap("groups");
    std::vector<int32_t> &dctx = ws.v_dctx, &qd = ws.v_qd;
    const bool ic = a.icBias != nullptr;
    const bool s2 = ic; // the context part as a slab of its own (distinct contexts x candidates)
    if (s2 && !ws.ctx_ready) distinct_contexts(plan.qc, dctx, qd); // (normally left there by rank_split_usable)
    if (!s2) {
        dctx.clear();
        qd.clear();
    }
    ws.ctx_ready = false;
    const int n_dc = (int)dctx.size();
    lap("contexts");
    a.kp1 = (a.k + 15) / 16 * 16; // (the item bias is not a column: the contraction's epilogue adds it -- k = 128 costs 8 steps of 16, not 9)
    a.kp2 = ic ? (a.n_conds + 15) / 16 * 16 : 16;
    a.nc = nc;
    a.nq = (int)nq;
    a.n_dctx = n_dc;
    auto up128 = [](int64_t v) { return (size_t)((v + 127) / 128 * 128); };
    // user groups per batch: an S1 slab of about 1 GiB (256-MiB batches were measured slower: 15.3 against 10.7 ms for the loop)
    int64_t bg = std::max<int64_t>(64, ((int64_t)1 << 30) / ((int64_t)nc * 4));
    bg = std::min<int64_t>(bg, ng);
    if (const char *e = getenv("CMI_RANK_BATCH")) bg = std::max<int64_t>(1, std::min<int64_t>(atoll(e), ng)); // tests: batching is invisible

    hipError_t e = hipSuccess;
    auto need = [&](RankWorkspace::Buf &b, size_t bytes, bool pinned = false) {
        if (e == hipSuccess) e = ws.need(b, bytes, pinned);
    };
    need(ws.dB, up128(nc) * a.kp1 * 4);
    need(ws.dcolc, up128(nc) * 4);
    // two streams: the selection of batch b runs beside the contraction of batch b + 1 (matrix pipe beside the L2 / memory pipes), each
    // batch on the slab / operand buffer of its parity.  CMI_RANK_ONE_STREAM=1: the round-4 form, one stream, one slab (A/B)
    const bool one_stream = getenv("CMI_RANK_ONE_STREAM") != nullptr; // (read per call: bench.py times the two kernels on their own with it)
    const bool two = !one_stream && ng > bg;
    // tile pruning (rank_topn_split_pruned): the contractions also write the rows' maxima over tiles of 64 candidates (0.8 % of the slab);
    // CMI_RANK_NO_PRUNE=1: the plain selection (tests compare the two forms entry for entry)
    const bool prune = getenv("CMI_RANK_NO_PRUNE") == nullptr;
    const size_t nt64 = (size_t)(nc + 63) / 64;
    need(ws.dA, up128(bg) * a.kp1 * 4);
    need(ws.dS, (size_t)bg * (size_t)nc * 4 + RANK_SLAB_SLACK);
    if (prune) need(ws.dM1, (size_t)bg * nt64 * 4);
    if (two) {
        need(ws.dAb, up128(bg) * a.kp1 * 4);
        need(ws.dSb, (size_t)bg * (size_t)nc * 4 + RANK_SLAB_SLACK);
        if (prune) need(ws.dM1b, (size_t)bg * nt64 * 4);
        // experiment builds, CMI_RANK_SEL_CUS=N: the selection's stream may only use N compute units of every XCD (of 32) and the
        // contraction runs on a stream of its own masked to the others, so that the two kernels overlap instead of taking turns
        // (docs/history/r06.md 3 has the sweep)
        const char *cus = cmi_exp_env("CMI_RANK_SEL_CUS");
        if (e == hipSuccess && !ws.sel_stream && cus && atoi(cus) > 0) {
            const int nsel = atoi(cus);
            int n_cu = 0;
            e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
            uint32_t msel[16] = {}, mgemm[16] = {};
            for (int c = 0; c < n_cu && c < 512; ++c) ((c / 8 < nsel) ? msel : mgemm)[c / 32] |= 1u << (c % 32); // bit c: XCD c % 8, slot c / 8
            if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&ws.sel_stream, (uint32_t)((n_cu + 31) / 32), msel);
            if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&ws.gemm_stream, (uint32_t)((n_cu + 31) / 32), mgemm);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ws.ev_gs, hipEventDisableTiming);
        }
        // experiment builds, CMI_RANK_SEL_PRIO=high|low: the selection's stream at the device's highest / lowest stream priority (the
        // instance's own stream, which carries the contraction, has the default one)
        if (const char *pr = cmi_exp_env("CMI_RANK_SEL_PRIO")) {
            int least = 0, greatest = 0;
            if (e == hipSuccess && !ws.sel_stream) e = hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (e == hipSuccess && !ws.sel_stream) e = hipStreamCreateWithPriority(&ws.sel_stream, hipStreamNonBlocking, !strcmp(pr, "high") ? greatest : least);
        }
        if (e == hipSuccess && !ws.sel_stream) e = hipStreamCreateWithFlags(&ws.sel_stream, hipStreamNonBlocking);
    }
    need(ws.dscr, std::max<size_t>(up128(bg), up128(n_dc)) * 4);
    need(ws.drc, (size_t)nq * 4);
    if (s2) {
        need(ws.dB2, up128(nc) * a.kp2 * 4);
        need(ws.dA2, up128(n_dc) * a.kp2 * 4);
        need(ws.dS2, (size_t)n_dc * (size_t)nc * 4 + RANK_SLAB_SLACK);
        if (prune) need(ws.dM2, (size_t)n_dc * nt64 * 4);
        need(ws.ddc, (size_t)n_dc * 4);
        need(ws.dqd, (size_t)nq * 4);
    }
    need(ws.dcand, (size_t)nc * 4);
    need(ws.dqu, (size_t)nq * 4);
    need(ws.dqc, (size_t)nq * 4);
    need(ws.dqg, (size_t)nq * 4);
    need(ws.dgu, (size_t)ng * 4);
    need(ws.dexptr, (size_t)(nq + 1) * 8);
    need(ws.dexcl, plan.excl_idx.size() * 4);
    need(ws.dtop, (size_t)nq * topn * 4);
    need(ws.dscore, (size_t)nq * topn * 8);
    need(ws.dcount, (size_t)nq * 4);
    need(ws.h_top, (size_t)nq * topn * 4, true);
    need(ws.h_score, (size_t)nq * topn * 8, true);
    need(ws.h_count, (size_t)nq * 4, true);
    hipEvent_t *evs[] = {&ws.ev0, &ws.ev1};
    for (hipEvent_t *ev : evs)
        if (e == hipSuccess && !*ev) e = hipEventCreate(ev);
    if (e != hipSuccess) return e;
    lap("buffers");
    auto up = [&](void *d, const void *s, size_t bytes) {
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, stream);
    };
    // the plan's index arrays: already on the device when this is the plan of the previous evaluation (ws.plan_valid: same tuples, by
    // content hash) and none of their buffers was re-allocated since
    const void *now[9] = {ws.dcand.p, ws.dqu.p, ws.dqc.p, ws.dqg.p, ws.dgu.p, ws.dexptr.p, ws.dexcl.p, s2 ? ws.ddc.p : nullptr, s2 ? ws.dqd.p : nullptr};
    bool resident = ws.plan_valid && ws.resident[0] != nullptr;
    for (int i = 0; i < 9 && resident; ++i) resident = ws.resident[i] == now[i];
    if (!resident) {
        up(ws.dcand.p, plan.cand.data(), (size_t)nc * 4);
        up(ws.dqu.p, plan.qu.data(), (size_t)nq * 4);
        up(ws.dqc.p, plan.qc.data(), (size_t)nq * 4);
        up(ws.dqg.p, qg.data(), (size_t)nq * 4);
        up(ws.dgu.p, gu.data(), (size_t)ng * 4);
        up(ws.dexptr.p, plan.excl_ptr.data(), (size_t)(nq + 1) * 8);
        up(ws.dexcl.p, plan.excl_idx.data(), plan.excl_idx.size() * 4);
        if (s2) {
            up(ws.ddc.p, dctx.data(), (size_t)n_dc * 4);
            up(ws.dqd.p, qd.data(), (size_t)nq * 4);
        }
        for (int i = 0; i < 9; ++i) ws.resident[i] = now[i];
    }
    lap("uploads");
    int32_t *dtop = (int32_t *)ws.dtop.p, *dcount = (int32_t *)ws.dcount.p;
    double *dscore = (double *)ws.dscore.p;
    if (e == hipSuccess) e = hipMemsetAsync(dtop, 0xff, (size_t)nq * topn * 4, stream);
    if (e == hipSuccess) e = hipMemsetAsync(dscore, 0, (size_t)nq * topn * 8, stream);
    if (e == hipSuccess) e = hipMemsetAsync(ws.dscr.p, 0, std::max<size_t>(up128(bg), up128(n_dc)) * 4, stream);
    a.cand = (const int32_t *)ws.dcand.p;
    a.qu = (const int32_t *)ws.dqu.p;
    a.qc = (const int32_t *)ws.dqc.p;
    a.dctx = (const int32_t *)ws.ddc.p;
    a.B1 = (float *)ws.dB.p;
    a.colc = (float *)ws.dcolc.p;
    a.B2 = (float *)ws.dB2.p;
    a.A2 = (float *)ws.dA2.p;
    a.rc = (float *)ws.drc.p;
    if (e == hipSuccess) e = rank_launch_split_operands(a, stream);
    if (e == hipSuccess) e = hipEventRecord(ws.ev0, stream);
    // S2: once per evaluation (row constant = the zeroed scratch)
    if (e == hipSuccess && s2)
        e = rank_launch_gemm<float>(a.A2, a.B2, (const float *)ws.dscr.p, (float *)ws.dS2.p, n_dc, nc, a.kp2, stream, nullptr, prune ? (float *)ws.dM2.p : nullptr);
    lap("memsets + operand launches");
    ws.host_ms[1] = ms_since(t_setup);
    const auto t_loop = std::chrono::steady_clock::now();
    std::vector<std::pair<int64_t, int64_t>> batches; // all enqueued first, consumed behind the device (see rank_run_device)
    const std::vector<int64_t> cuts = batch_cuts(ng, bg);
    auto ev_at = [&](std::vector<hipEvent_t> &v, size_t i) -> hipEvent_t {
        while (v.size() <= i && e == hipSuccess) {
            hipEvent_t ev = nullptr;
            e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) v.push_back(ev);
        }
        return i < v.size() ? v[i] : nullptr;
    };
    hipStream_t sel = two ? ws.sel_stream : stream;
    // (the contraction of batch b + 1 made to wait for the selection of batch b -- the kernels never side by side, only the lists' copies
    // overlapped -- measured the same 6.3 ms as the overlapped form: docs/history/r06.md 3)
    hipStream_t gs = two && ws.gemm_stream ? ws.gemm_stream : stream; // (the contraction's stream: the instance's own, except in the CU-split experiment)
    if (gs != stream && e == hipSuccess) {
        e = hipEventRecord(ws.ev_gs, stream); // behind the operands and the S2 contraction
        if (e == hipSuccess) e = hipStreamWaitEvent(gs, ws.ev_gs, 0);
    }
    for (size_t b = 0; b + 1 < cuts.size() && e == hipSuccess; ++b) {
        const int64_t g0 = cuts[b];
        const int n = (int)(cuts[b + 1] - g0);
        const int64_t q0 = gq0[(size_t)g0], q1 = gq0[(size_t)(g0 + n)];
        float *dAx = (float *)((two && (b & 1)) ? ws.dAb.p : ws.dA.p), *dSx = (float *)((two && (b & 1)) ? ws.dSb.p : ws.dS.p);
        float *dMx = prune ? (float *)((two && (b & 1)) ? ws.dM1b.p : ws.dM1.p) : nullptr;
        // the slab of this parity is free once the selection of batch b - 2 has read it
        if (two && b >= 2 && e == hipSuccess) e = hipStreamWaitEvent(gs, ev_at(ws.evsel, b - 2), 0);
        // (the builder leaves its per-row constant -- zero here -- in the scratch the contraction then reads as its row constant)
        if (e == hipSuccess) e = rank_launch_split_users(a, (const int32_t *)ws.dgu.p + g0, n, dAx, (float *)ws.dscr.p, gs);
        if (e == hipSuccess) e = ws.kernel_event(4 * b, gs);
        if (e == hipSuccess) e = rank_launch_gemm<float>(dAx, a.B1, (const float *)ws.dscr.p, dSx, n, nc, a.kp1, gs, a.colc, dMx);
        if (e == hipSuccess) e = ws.kernel_event(4 * b + 1, gs);
        if (two && e == hipSuccess) {
            hipEvent_t g = ev_at(ws.evgemm, b);
            if (e == hipSuccess) e = hipEventRecord(g, gs);
            if (e == hipSuccess) e = hipStreamWaitEvent(sel, g, 0);
        }
        if (e == hipSuccess) e = ws.kernel_event(4 * b + 2, sel);
        if (e == hipSuccess)
            e = rank_launch_split_select(dSx, s2 ? (const float *)ws.dS2.p : nullptr, a, (const int32_t *)ws.dqg.p,
                                         (const int32_t *)ws.dqd.p, (int)g0, (int)q0, (int)(q1 - q0), (const int64_t *)ws.dexptr.p,
                                         (const int32_t *)ws.dexcl.p, thold, topn, dtop, dscore, dcount, sel, dMx, prune && s2 ? (const float *)ws.dM2.p : nullptr);
        if (e == hipSuccess) e = ws.kernel_event(4 * b + 3, sel);
        if (two && e == hipSuccess) {
            hipEvent_t sd = ev_at(ws.evsel, b);
            if (e == hipSuccess) e = hipEventRecord(sd, sel);
        }
        const size_t nqb = (size_t)(q1 - q0);
        if (e == hipSuccess)
            e = hipMemcpyAsync((int32_t *)ws.h_top.p + (size_t)q0 * topn, dtop + (size_t)q0 * topn, nqb * topn * 4, hipMemcpyDeviceToHost, sel);
        if (e == hipSuccess)
            e = hipMemcpyAsync((double *)ws.h_score.p + (size_t)q0 * topn, dscore + (size_t)q0 * topn, nqb * topn * 8, hipMemcpyDeviceToHost, sel);
        if (e == hipSuccess) e = hipMemcpyAsync((int32_t *)ws.h_count.p + q0, dcount + q0, nqb * 4, hipMemcpyDeviceToHost, sel);
        if (e == hipSuccess) e = ws.batch_event(batches.size(), sel);
        batches.emplace_back(q0, q1);
    }
    // the loop ends when the last batch's lists are on the host: ev1 on the main stream, behind the selection stream's last event
    if (two && e == hipSuccess && !batches.empty()) e = hipStreamWaitEvent(stream, ws.evb[batches.size() - 1], 0);
    if (e == hipSuccess) e = hipEventRecord(ws.ev1, stream);
    e = ws.consume_batches(e, batches, on_batch, t_loop);
    if (e == hipSuccess && ms) e = hipEventElapsedTime(ms, ws.ev0, ws.ev1);
    for (size_t b = 0; b < batches.size() && e == hipSuccess; ++b) { // (with two streams the two kernels' times overlap: their sum exceeds the loop)
        float g = 0.f, t = 0.f;
        e = hipEventElapsedTime(&g, ws.evk[4 * b], ws.evk[4 * b + 1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, ws.evk[4 * b + 2], ws.evk[4 * b + 3]);
        ws.kernel_ms[0] += g;
        ws.kernel_ms[1] += t;
    }
    // flops of THIS form: the two contractions it actually runs
    if (flops) *flops = 2.0 * (double)ng * (double)nc * (double)a.kp1 + (s2 ? 2.0 * (double)n_dc * (double)nc * (double)a.kp2 : 0.0);
    return e;
}

template hipError_t rank_run_device<float>(hipStream_t, RankWorkspace &, const RankPlan &, const RankOperands<float> &, double, int,
                                           const std::function<void(int64_t, int64_t)> &, float *, double *);
template hipError_t rank_run_device<double>(hipStream_t, RankWorkspace &, const RankPlan &, const RankOperands<double> &, double, int,
                                            const std::function<void(int64_t, int64_t)> &, float *, double *);

} // namespace cmi

// operand builders of the MF family (BiasedMF, PMF, CAMF_*): see rank_kernels.hip
template <typename T>
static RankOperands<T> mf_operands(cmi_instance *h, int k_logical, bool contextual, bool ic_used) {
    RankOperands<T> ops;
    ops.k_logical = k_logical;
    ops.build_items = [h](T *dB, const int32_t *dcand, int nc, int kp, hipStream_t s) {
        RankItemsArgs<T> ia{(const T *)h->state[CMI_STATE_Q], (const T *)h->state[CMI_STATE_ITEM_BIAS],
                            (const T *)h->state[CMI_STATE_IC_BIAS], dcand, dB, nc, h->k, kp, h->n_conds};
        return rank_launch_build_items<T>(ia, s);
    };
    ops.build_queries = [h, contextual, ic_used](T *dA, T *drc, const int32_t *dqu, const int32_t *dqc, int n, int kp, hipStream_t s) {
        RankQueryArgs<T> qa{(const T *)h->state[CMI_STATE_P],
                            (const T *)h->state[CMI_STATE_USER_BIAS],
                            (const T *)h->state[CMI_STATE_UC_BIAS],
                            (const T *)h->state[CMI_STATE_COND_BIAS],
                            dqu,
                            dqc,
                            contextual ? h->d_ctx_ptr : nullptr,
                            contextual ? h->d_ctx_conds : nullptr,
                            dA,
                            drc,
                            h->hp.gm,
                            h->k,
                            kp,
                            h->n_conds,
                            ic_used ? 1 : 0};
        return rank_launch_build_queries<T>(qa, n, s);
    };
    return ops;
}

// operand builders of SVD++ / CAMF_ICS / CAMF_LCS / CAMF_MCS: see ext_kernels.hip
template <typename T>
static RankOperands<T> ext_operands(cmi_instance *h, int k_logical) {
    RankOperands<T> ops;
    ops.k_logical = k_logical;
    ops.build_items = [h](T *dB, const int32_t *dcand, int nc, int kp, hipStream_t s) {
        return launch_ext_rank_items<T>(cmi_ext_eval_args<T>(h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 1), dcand, nc, dB, kp, s);
    };
    ops.build_queries = [h](T *dA, T *drc, const int32_t *dqu, const int32_t *dqc, int n, int kp, hipStream_t s) {
        return launch_ext_rank_queries<T>(cmi_ext_eval_args<T>(h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 1), dqu, dqc, n, dA, drc, kp, s);
    };
    return ops;
}

extern "C" int cmi_last_rank_ms(cmi_handle h, float *ms, double *flops) {
    if (!h) return CMI_E_INVALID;
    if (ms) *ms = h->last_rank_ms;
    if (flops) *flops = h->last_rank_flops;
    return CMI_OK;
}

extern "C" int cmi_last_rank_kernel_ms(cmi_handle h, double out[2]) {
    if (!h || !out) return CMI_E_INVALID;
    out[0] = h->rank_ws.kernel_ms[0];
    out[1] = h->rank_ws.kernel_ms[1];
    return CMI_OK;
}

extern "C" int cmi_last_rank_host_ms(cmi_handle h, double out[5]) {
    if (!h || !out) return CMI_E_INVALID;
    for (int i = 0; i < 5; ++i) out[i] = h->rank_ws.host_ms[i];
    return CMI_OK;
}

// host-only: the bookkeeping of cmi_eval_rankings without a device -- candidates (in HashSet<Integer> order, minus the ignored),
// queries, their correct items and the candidate positions excluded per query.  Two-call protocol: sizes first (null
// outputs), then the arrays.  sizes = {n_cand, n_queries, n_truth, n_excl}.
extern "C" int cmi_rank_plan(int32_t n_users, int32_t n_items, int64_t n_train, const int32_t *tu, const int32_t *tj,
                             const int32_t *tctx, const double *tr, int64_t n_test, const int32_t *su, const int32_t *sj,
                             const int32_t *sctx, const double *sr, double bin_thold, int num_ignore, int64_t sizes[4],
                             int32_t *cand, int32_t *q_user, int32_t *q_ctx, int64_t *truth_ptr, int32_t *truth_items,
                             int64_t *excl_ptr, int32_t *excl_idx) {
    if (!sizes || n_users <= 0 || n_items <= 0 || n_train < 0 || n_test < 0 || (n_train > 0 && (!tu || !tj || !tctx)) ||
        (n_test > 0 && (!su || !sj || !sctx || !sr)))
        return CMI_E_INVALID;
    for (int64_t t = 0; t < n_train; ++t)
        if (tu[t] < 0 || tu[t] >= n_users || tj[t] < 0 || tj[t] >= n_items || tctx[t] < 0) return CMI_E_INVALID;
    for (int64_t t = 0; t < n_test; ++t)
        if (su[t] < 0 || su[t] >= n_users || sj[t] < 0 || sj[t] >= n_items || sctx[t] < 0) return CMI_E_INVALID;
    RankPlan plan;
    rank_build_plan(n_users, n_items, RankTuples{n_train, tu, tj, tctx, tr}, RankTuples{n_test, su, sj, sctx, sr}, bin_thold,
                    num_ignore, plan);
    sizes[0] = (int64_t)plan.cand.size();
    sizes[1] = (int64_t)plan.qu.size();
    sizes[2] = (int64_t)plan.truth_items.size();
    sizes[3] = (int64_t)plan.excl_idx.size();
    if (cand) std::copy(plan.cand.begin(), plan.cand.end(), cand);
    if (q_user) std::copy(plan.qu.begin(), plan.qu.end(), q_user);
    if (q_ctx) std::copy(plan.qc.begin(), plan.qc.end(), q_ctx);
    if (truth_ptr) std::copy(plan.truth_ptr.begin(), plan.truth_ptr.end(), truth_ptr);
    if (truth_items) std::copy(plan.truth_items.begin(), plan.truth_items.end(), truth_items);
    if (excl_ptr) std::copy(plan.excl_ptr.begin(), plan.excl_ptr.end(), excl_ptr);
    if (excl_idx) std::copy(plan.excl_idx.begin(), plan.excl_idx.end(), excl_idx);
    return CMI_OK;
}

// host-only: the 18 measures of one ranked list, as cmi_eval_rankings computes them per query
extern "C" int cmi_rank_list_measures(const int32_t *ranked, int len, const int32_t *truth_sorted, int n_truth, int num_dropped,
                                      int num_recs, double out[18]) {
    if (len < 0 || n_truth <= 0 || num_recs < 1 || !truth_sorted || !out || (len > 0 && !ranked)) return CMI_E_INVALID;
    for (int i = 1; i < n_truth; ++i)
        if (truth_sorted[i - 1] >= truth_sorted[i]) return CMI_E_INVALID; // strictly ascending (binary search)
    const Truth t{truth_sorted, n_truth};
    list_measures(ranked, len, t, num_dropped, num_recs, out);
    return CMI_OK;
}

extern "C" int cmi_java_int_hashset_order(int64_t n, const int32_t *values, int32_t *out, int64_t *n_out) {
    if (n < 0 || (n > 0 && (!values || !out)) || !n_out) return CMI_E_INVALID;
    std::vector<int32_t> first;
    {
        std::vector<int32_t> sorted(values, values + n);
        std::sort(sorted.begin(), sorted.end());
        sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
        std::vector<char> seen(sorted.size(), 0);
        for (int64_t i = 0; i < n; ++i) {
            const size_t p = std::lower_bound(sorted.begin(), sorted.end(), values[i]) - sorted.begin();
            if (!seen[p]) {
                seen[p] = 1;
                first.push_back(values[i]);
            }
        }
    }
    const std::vector<int32_t> ord = java_int_hashset_order(first);
    std::copy(ord.begin(), ord.end(), out);
    *n_out = (int64_t)ord.size();
    return CMI_OK;
}

static int eval_rankings_impl(cmi_handle h, int64_t n_train, const int32_t *tu, const int32_t *tj,
                                 const int32_t *tctx, const double *tr, int64_t n_test, const int32_t *su,
                                 const int32_t *sj, const int32_t *sctx, const double *sr, double bin_thold,
                                 int num_recs, int num_ignore, int strategy, double out[CMI_RANK_MEASURES],
                                 int64_t *n_queries, int32_t *q_user, int32_t *q_ctx, int32_t *q_count,
                                 int32_t *top_items, double *top_scores);
extern "C" int cmi_eval_rankings(cmi_handle h, int64_t n_train, const int32_t *tu, const int32_t *tj,
                                 const int32_t *tctx, const double *tr, int64_t n_test, const int32_t *su,
                                 const int32_t *sj, const int32_t *sctx, const double *sr, double bin_thold,
                                 int num_recs, int num_ignore, int strategy, double out[CMI_RANK_MEASURES],
                                 int64_t *n_queries, int32_t *q_user, int32_t *q_ctx, int32_t *q_count,
                                 int32_t *top_items, double *top_scores) {
    if (!h) return CMI_E_INVALID;
    try { // exception barrier: the plan and the measures allocate per-range vectors on the host pool
        return eval_rankings_impl(h, n_train, tu, tj, tctx, tr, n_test, su, sj, sctx, sr, bin_thold, num_recs, num_ignore, strategy, out, n_queries,
                                  q_user, q_ctx, q_count, top_items, top_scores);
    } catch (const std::exception &e) {
        CMI_FAIL(h, CMI_E_HOST, "eval_rankings: host-side failure: %s", e.what());
    } catch (...) {
        CMI_FAIL(h, CMI_E_HOST, "eval_rankings: host-side failure (unknown exception)");
    }
}

static int eval_rankings_impl(cmi_handle h, int64_t n_train, const int32_t *tu, const int32_t *tj,
                                 const int32_t *tctx, const double *tr, int64_t n_test, const int32_t *su,
                                 const int32_t *sj, const int32_t *sctx, const double *sr, double bin_thold,
                                 int num_recs, int num_ignore, int strategy, double out[CMI_RANK_MEASURES],
                                 int64_t *n_queries, int32_t *q_user, int32_t *q_ctx, int32_t *q_count,
                                 int32_t *top_items, double *top_scores) {
    if (!h) return CMI_E_INVALID;
    if (!out) CMI_FAIL(h, CMI_E_INVALID, "eval_rankings: null output");
    if (int rc = cmi_sync_table_from_arena(h)) return rc;
    if (n_train < 0 || n_test < 0 || (n_train > 0 && (!tu || !tj || !tctx)) || (n_test > 0 && (!su || !sj || !sctx || !sr)))
        CMI_FAIL(h, CMI_E_INVALID, "eval_rankings: null tuple arrays");
    if (num_recs < 1)
        CMI_FAIL(h, CMI_E_INVALID,
                 "eval_rankings: -topN must be >= 1 (with -topN <= 0 the reference's cut-off list holds a non-positive n: "
                 "carskit/eval/Measures.java:13-16 throws for n<0)");
    if (strategy != CMI_RANK_UCU && strategy != CMI_RANK_UC) CMI_FAIL(h, CMI_E_INVALID, "eval_rankings: strategy must be CMI_RANK_UCU or CMI_RANK_UC");
    const bool ext = h->model >= CMI_MODEL_SVDPP && h->model <= CMI_MODEL_CAMF_MCS;
    const bool contextual = h->model != CMI_MODEL_BIASEDMF && h->model != CMI_MODEL_PMF && h->model != CMI_MODEL_SVDPP;
    if ((contextual || ext) && !h->have_ratings)
        CMI_FAIL(h, CMI_E_INVALID, "eval_rankings: the context table comes from cmi_set_ratings; call it first");
    auto check = [&](int64_t n, const int32_t *u, const int32_t *j, const int32_t *c, const char *what) -> int {
        const int nt = host_threads(n);
        std::vector<int64_t> bad((size_t)nt, -1); // first offending tuple of every range
        parallel_ranges(n, nt, [&](int part, int64_t b, int64_t e) {
            for (int64_t t = b; t < e; ++t)
                if (u[t] < 0 || u[t] >= h->n_users || j[t] < 0 || j[t] >= h->n_items || c[t] < 0 || (contextual && c[t] >= h->n_ctx)) {
                    bad[(size_t)part] = t;
                    return;
                }
        });
        for (int64_t t : bad) {
            if (t < 0) continue;
            if (u[t] < 0 || u[t] >= h->n_users || j[t] < 0 || j[t] >= h->n_items)
                CMI_FAIL(h, CMI_E_INVALID, "eval_rankings: %s user/item id out of range at tuple %lld", what, (long long)t);
            CMI_FAIL(h, CMI_E_INVALID, "eval_rankings: %s context id %d out of range at tuple %lld", what, c[t], (long long)t);
        }
        return CMI_OK;
    };
    if (int rc = check(n_train, tu, tj, tctx, "train")) return rc;
    if (int rc = check(n_test, su, sj, sctx, "test")) return rc;
    CMI_HIP(h, hipSetDevice(h->device));
    if (n_queries) *n_queries = 0;

    const auto t_all = std::chrono::steady_clock::now();
    RankWorkspace &ws = h->rank_ws;
    RankPlan &plan = ws.plan; // its arrays keep their capacity between evaluations
    // the same tuples as the previous evaluation (an early-stop loop evaluates after every epoch): the plan is kept.  Identity = sizes +
    // a 64-bit hash of the arrays' CONTENT (ranges on the host's cores: half a millisecond for 2.5 M tuples against 4 ms for the plan)
    RankWorkspace::PlanKey key;
    key.n_train = n_train, key.n_test = n_test, key.n_users = h->n_users, key.n_items = h->n_items, key.bin_thold = bin_thold, key.num_ignore = num_ignore;
    {
        struct Arr {
            const void *p;
            size_t bytes;
        } arrs[] = {{tu, (size_t)n_train * 4}, {tj, (size_t)n_train * 4}, {tctx, (size_t)n_train * 4}, {tr, tr ? (size_t)n_train * 8 : 0},
                    {su, (size_t)n_test * 4},  {sj, (size_t)n_test * 4},  {sctx, (size_t)n_test * 4},  {sr, (size_t)n_test * 8}};
        // one pass of the host's cores over all eight arrays as one run of 32-bit words; every thread keeps four independent chains
        // (the multiply's latency is the pace of one)
        uint64_t hsh = 0x9E3779B97F4A7C15ull;
        int64_t first[9] = {0};
        for (int i = 0; i < 8; ++i) first[i + 1] = first[i] + (int64_t)(arrs[i].bytes / 4);
        const int64_t words = first[8];
        const int nt = host_threads(words / 16);
        std::vector<uint64_t> part((size_t)nt, 0);
        parallel_ranges(words, nt, [&](int t, int64_t b, int64_t e) {
            uint64_t x[4] = {0xCBF29CE484222325ull ^ (uint64_t)b, 0x84222325CBF29CE4ull, 0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full};
            int ai = 0;
            while (ai < 7 && first[ai + 1] <= b) ++ai;
            for (int64_t i = b; i < e;) {
                while (first[ai + 1] <= i) ++ai;
                const uint32_t *w = (const uint32_t *)arrs[ai].p - first[ai];
                const int64_t stop = std::min<int64_t>(e, first[ai + 1]);
                for (; i + 4 <= stop; i += 4)
                    for (int l = 0; l < 4; ++l) {
                        x[l] ^= w[i + l];
                        x[l] *= 0x100000001B3ull;
                        x[l] ^= x[l] >> 29;
                    }
                for (; i < stop; ++i) {
                    x[0] ^= w[i];
                    x[0] *= 0x100000001B3ull;
                    x[0] ^= x[0] >> 29;
                }
            }
            part[(size_t)t] = ((x[0] * 0xFF51AFD7ED558CCDull ^ x[1]) * 0xFF51AFD7ED558CCDull ^ x[2]) * 0xFF51AFD7ED558CCDull ^ x[3];
        });
        for (uint64_t x : part) hsh = (hsh ^ x) * 0xFF51AFD7ED558CCDull + (hsh >> 31);
        for (const Arr &a : arrs) hsh = (hsh ^ (a.bytes + (a.p ? 1 : 0))) * 0x100000001B3ull;
        key.hash = hsh;
    }
    if (!(ws.plan_valid && ws.plan_key == key) || cmi_exp_env("CMI_RANK_NO_PLAN_CACHE")) {
        ws.plan_valid = false;
        rank_build_plan(h->n_users, h->n_items, RankTuples{n_train, tu, tj, tctx, tr}, RankTuples{n_test, su, sj, sctx, sr}, bin_thold,
                        num_ignore, plan);
        ws.plan_key = key;
        for (const void *&r : ws.resident) r = nullptr;
        ws.plan_valid = true;
    }
    ws.host_ms[0] = ms_since(t_all);
    ws.host_ms[1] = ws.host_ms[2] = ws.host_ms[3] = 0.0;
    ws.kernel_ms[0] = ws.kernel_ms[1] = 0.0;
    const int64_t nq = (int64_t)plan.qu.size();
    double *vals = ws.vals.need((size_t)nq * N_MEAS + 1); // only the rows of queries with a list are written and read
    std::vector<int32_t> no_lists;
    const int32_t *top_count = nullptr;
    RankFolded folded;  // ucu: queries / users whose means are already taken
    double *umeans = ws.umeans.need((size_t)std::min<int64_t>(nq, h->n_users) * N_MEAS + 1);
    if (nq > 0 && !plan.cand.empty()) {
        const bool ic_used = h->state[CMI_STATE_IC_BIAS] != nullptr;
        const int k_logical = ext ? h->k + (h->model == CMI_MODEL_SVDPP ? 1 : 0)
                                  : h->k + 1 + (ic_used ? h->n_conds : 0); // [factors | 1 or itemBias | one-hot conditions or icBias row]
        const bool times = getenv("CMI_PLAN_TIMES") != nullptr;
        auto on_batch = [&](int64_t q0, int64_t q1) {
            const auto tb = std::chrono::steady_clock::now();
            struct Lap {
                bool on;
                std::chrono::steady_clock::time_point t0;
                int64_t n;
                ~Lap() {
                    if (on) fprintf(stderr, "rank batch of %lld queries: measures + user means %.3f ms on the host\n", (long long)n, ms_since(t0));
                }
            } lap{times, tb, q1 - q0};
            rank_measures_range(plan, num_recs, (const int32_t *)ws.h_top.p, (const double *)ws.h_score.p, (const int32_t *)ws.h_count.p, q0, q1,
                                vals, q_user, q_ctx, q_count, top_items, top_scores);
            // the batches arrive in query order: the users they complete are averaged here, behind the device
            if (strategy == CMI_RANK_UCU) rank_fold_users(plan, (const int32_t *)ws.h_count.p, vals, umeans, folded, q1, q1 >= nq);
            else rank_sum_queries((const int32_t *)ws.h_count.p, vals, folded, q1); // uc: the serial sum over the queries, behind the device
        };
        hipError_t e;
        if (ext && h->f64) e = rank_run_device<double>(h->stream, ws, plan, ext_operands<double>(h, k_logical), bin_thold, num_recs, on_batch,
                                                       &h->last_rank_ms, &h->last_rank_flops);
        else if (ext) e = rank_run_device<float>(h->stream, ws, plan, ext_operands<float>(h, k_logical), bin_thold, num_recs, on_batch,
                                                 &h->last_rank_ms, &h->last_rank_flops);
        else if (h->f64) e = rank_run_device<double>(h->stream, ws, plan, mf_operands<double>(h, k_logical, contextual, ic_used), bin_thold,
                                                     num_recs, on_batch, &h->last_rank_ms, &h->last_rank_flops);
        else if (rank_split_usable(plan, num_recs, ws)) {
            RankSplitArgs sa{};
            sa.P = (const float *)h->state[CMI_STATE_P];
            sa.Q = (const float *)h->state[CMI_STATE_Q];
            sa.userBias = (const float *)h->state[CMI_STATE_USER_BIAS];
            sa.itemBias = (const float *)h->state[CMI_STATE_ITEM_BIAS];
            sa.ucBias = (const float *)h->state[CMI_STATE_UC_BIAS];
            sa.icBias = ic_used ? (const float *)h->state[CMI_STATE_IC_BIAS] : nullptr;
            sa.condBias = (const float *)h->state[CMI_STATE_COND_BIAS];
            sa.ctx_ptr = contextual ? h->d_ctx_ptr : nullptr;
            sa.ctx_conds = contextual ? h->d_ctx_conds : nullptr;
            sa.gm = h->hp.gm;
            sa.k = h->k;
            sa.n_conds = h->n_conds;
            e = rank_run_device_split(h->stream, ws, plan, sa, bin_thold, num_recs, on_batch, &h->last_rank_ms, &h->last_rank_flops);
        } else e = rank_run_device<float>(h->stream, ws, plan, mf_operands<float>(h, k_logical, contextual, ic_used), bin_thold, num_recs,
                                          on_batch, &h->last_rank_ms, &h->last_rank_flops);
        CMI_HIP(h, e);
        top_count = (const int32_t *)ws.h_count.p;
    } else {
        no_lists.assign((size_t)nq, 0);
        top_count = no_lists.data();
        rank_measures_range(plan, num_recs, nullptr, nullptr, top_count, 0, nq, vals, q_user, q_ctx, q_count, top_items, top_scores);
    }
    const auto t_avg = std::chrono::steady_clock::now();
    rank_average(plan, strategy, top_count, vals, umeans, folded, out);
    if (getenv("CMI_PLAN_TIMES")) fprintf(stderr, "rank last batch's measures %.3f ms, averages %.3f ms\n", ws.host_ms[3], ms_since(t_avg));
    ws.host_ms[3] += ms_since(t_avg);
    ws.host_ms[4] = ms_since(t_all);
    if (n_queries) *n_queries = (int64_t)plan.qu.size();
    return CMI_OK;
}

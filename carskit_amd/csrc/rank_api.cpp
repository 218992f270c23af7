// rank_api.cpp -- cmi_eval_rankings: the reference's top-N evaluation (Recommender.evalRankings,
// src/carskit/generic/Recommender.java:668-964) with the O(queries x items x k) scoring on the GPU.
// Host side: query / candidate / exclusion bookkeeping and the (tiny, per-query) metric formulas; device side:
// rank_kernels.hip.  No CPU scoring path exists.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "cmi_instance.hpp"
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

int hits_at(const int32_t *ranked, int len, const Truth &t, int n) { // Measures.HitsAt over the FULL ranked list
    int hits = 0;
    for (int i = 0; i < len; ++i)
        if (t.has(ranked[i])) {
            if (i >= n) break;
            ++hits;
        }
    return hits;
}

double auc(const int32_t *ranked, int len, const Truth &t, int num_dropped) {
    int num_rele = 0; // Lists.overlapSize(groundTruth, rankedList)
    for (int i = 0; i < len; ++i) num_rele += t.has(ranked[i]);
    const long num_eval_items = (long)len + num_dropped;
    const long num_eval_pairs = (num_eval_items - num_rele) * num_rele;
    if (num_eval_pairs == 0) return 0.5;
    long correct = 0, hits = 0;
    for (int i = 0; i < len; ++i) {
        if (!t.has(ranked[i])) correct += hits;
        else ++hits;
    }
    const long num_miss = t.n - num_rele; // Lists.exceptSize(groundTruth, rankedList)
    correct += hits * ((long)num_dropped - num_miss);
    return (double)correct / (double)num_eval_pairs;
}

double ap(const int32_t *ranked, int len, const Truth &t) {
    int hits = 0;
    double s = 0.0;
    for (int i = 0; i < len; ++i)
        if (t.has(ranked[i])) {
            ++hits;
            s += hits / (i + 1.0);
        }
    return hits > 0 ? s / t.n : 0.0;
}

inline double log2j(double x) { return std::log(x) / std::log(2.0); } // Maths.log(x, 2)

double ndcg(const int32_t *ranked, int len, const Truth &t) {
    double dcg = 0.0, idcg = 0.0;
    for (int i = 0; i < len; ++i)
        if (t.has(ranked[i])) dcg += 1.0 / log2j(i + 2);
    for (int i = 0; i < t.n; ++i) idcg += 1.0 / log2j(i + 2);
    return dcg / idcg;
}

double rr(const int32_t *ranked, int len, const Truth &t) {
    for (int i = 0; i < len; ++i)
        if (t.has(ranked[i])) return 1.0 / (i + 1.0);
    return 0.0;
}

struct NanMean { // happy.coding.math.Stats.mean(Collection): NaN entries are skipped; empty -> 0/0 = NaN
    double s = 0.0;
    long c = 0;
    void add(double x) {
        if (!std::isnan(x)) {
            s += x;
            ++c;
        }
    }
    double value() const { return c ? s / (double)c : std::nan(""); }
};

// tuple indices `sel` ordered by (user, context, item): counting sort by user, then each user's short run is sorted
// (O(n) instead of one comparison sort over millions of tuples)
std::vector<int64_t> order_by_user_ctx_item(int n_users, const std::vector<int64_t> &sel, const int32_t *u, const int32_t *c,
                                            const int32_t *j) {
    std::vector<int64_t> off((size_t)n_users + 1, 0), out(sel.size());
    for (int64_t t : sel) off[(size_t)u[t] + 1]++;
    for (int l = 0; l < n_users; ++l) off[(size_t)l + 1] += off[(size_t)l];
    {
        std::vector<int64_t> cur(off.begin(), off.end() - 1);
        for (int64_t t : sel) out[(size_t)cur[(size_t)u[t]]++] = t;
    }
    for (int l = 0; l < n_users; ++l)
        if (off[(size_t)l + 1] - off[(size_t)l] > 1)
            std::sort(out.begin() + off[(size_t)l], out.begin() + off[(size_t)l + 1], [&](int64_t a, int64_t b) {
                if (c[a] != c[b]) return c[a] < c[b];
                return j[a] < j[b];
            });
    return out;
}

constexpr int N_MEAS = 18; // Pre,Rec,AUC,MAP,NDCG,MRR x {5,10,N}

// the 18 measures of ONE ranked list (already cut at num_recs): index = measure * 3 + cut-off, cut-offs {5, 10, num_recs}
// (Recommender.java:852-858 through carskit.eval.Measures.*At)
void list_measures(const int32_t *ranked, int len, const Truth &t, int num_dropped, int num_recs, double vals[N_MEAS]) {
    const int cut[3] = {5, 10, num_recs};
    for (int c = 0; c < 3; ++c) {
        const int n = cut[c], tl = std::min(n, len);
        const int hits = hits_at(ranked, len, t, n);
        vals[0 + c] = hits / (n + 0.0);
        vals[3 + c] = hits / (t.n + 0.0);
        vals[6 + c] = auc(ranked, tl, t, num_dropped);
        vals[9 + c] = ap(ranked, tl, t);
        vals[12 + c] = ndcg(ranked, tl, t);
        vals[15 + c] = rr(ranked, tl, t);
    }
}

} // namespace

namespace cmi {

void rank_build_plan(int n_users, int n_items, const RankTuples &train, const RankTuples &test, double bin_thold,
                     int num_ignore, RankPlan &plan) {
    // candidate items: rateDao.getItemList(trainMatrix) -> HashSet<Integer> (DataDAO.java:1210-1218)
    std::vector<int32_t> first_seen, degree(n_items, 0);
    for (int64_t t = 0; t < train.n; ++t) {
        if (train.r && train.r[t] == 0.0) continue; // a sparse matrix holds no zero entries
        if (degree[train.j[t]]++ == 0) first_seen.push_back(train.j[t]);
    }
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

    // queries: test positives (rate > threshold) grouped by (user, context)  (DataDAO.getUserCtxList, DataDAO.java:1114-1140)
    std::vector<int64_t> pos;
    for (int64_t t = 0; t < test.n; ++t)
        if (test.r[t] != 0.0 && test.r[t] > bin_thold) pos.push_back(t);
    pos = order_by_user_ctx_item(n_users, pos, test.u, test.ctx, test.j);
    std::vector<int32_t> &qu = plan.qu, &qc = plan.qc, &truth_items = plan.truth_items;
    std::vector<int64_t> &truth_ptr = plan.truth_ptr;
    qu.clear();
    qc.clear();
    truth_items.clear();
    truth_ptr.assign(1, 0);
    for (size_t i = 0; i < pos.size();) {
        size_t e = i;
        const size_t before = truth_items.size();
        while (e < pos.size() && test.u[pos[e]] == test.u[pos[i]] && test.ctx[pos[e]] == test.ctx[pos[i]]) {
            const int32_t j = test.j[pos[e]];
            if (cand_pos[j] >= 0 && (truth_items.size() == before || truth_items.back() != j)) truth_items.push_back(j);
            ++e;
        }
        if (truth_items.size() > before) { // correctItems non-empty (Recommender.java:789-790)
            qu.push_back(test.u[pos[i]]);
            qc.push_back(test.ctx[pos[i]]);
            truth_ptr.push_back((int64_t)truth_items.size());
        }
        i = e;
    }
    const int64_t nq = (int64_t)qu.size();

    // exclusions: items the user rated in the same context in the training set (Recommender.java:793, 814-816)
    std::vector<int64_t> tord;
    for (int64_t t = 0; t < train.n; ++t)
        if (!(train.r && train.r[t] == 0.0)) tord.push_back(t);
    tord = order_by_user_ctx_item(n_users, tord, train.u, train.ctx, train.j);
    std::vector<int64_t> &excl_ptr = plan.excl_ptr;
    std::vector<int32_t> &excl_idx = plan.excl_idx;
    excl_ptr.assign(1, 0);
    excl_idx.clear();
    {
        size_t p = 0;
        for (int64_t q = 0; q < nq; ++q) {
            while (p < tord.size() && (train.u[tord[p]] < qu[q] || (train.u[tord[p]] == qu[q] && train.ctx[tord[p]] < qc[q]))) ++p;
            size_t e = p;
            while (e < tord.size() && train.u[tord[e]] == qu[q] && train.ctx[tord[e]] == qc[q]) {
                const int32_t cp = cand_pos[train.j[tord[e]]];
                if (cp >= 0 && (excl_idx.size() == (size_t)excl_ptr.back() || excl_idx.back() != cp)) excl_idx.push_back(cp);
                ++e;
            }
            excl_ptr.push_back((int64_t)excl_idx.size());
        }
    }

}

void rank_metrics(const RankPlan &plan, int strategy, int num_recs, const std::vector<int32_t> &top_idx,
                  const std::vector<double> &top_score, const std::vector<int32_t> &top_count, double *out,
                  int32_t *q_user, int32_t *q_ctx, int32_t *q_count, int32_t *top_items, double *top_scores) {
    const int64_t nq = (int64_t)plan.qu.size();
    const int nc = (int)plan.cand.size();
    for (int m = 0; m < CMI_RANK_MEASURES; ++m) out[m] = std::nan("");
    out[18] = out[19] = out[20] = 0.0; // D5/D10/DN: isDiverseUsed=false (Recommender.java:939-941)
    // metrics, per query then averaged per strategy (Recommender.java:850-960)
    NanMean total[N_MEAS], per_user[N_MEAS];
    std::vector<int32_t> ranked(num_recs);
    auto flush_user = [&]() {
        for (int m = 0; m < N_MEAS; ++m) {
            total[m].add(per_user[m].value());
            per_user[m] = NanMean();
        }
    };
    int64_t emitted = 0;
    for (int64_t q = 0; q < nq; ++q) {
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
            const int num_dropped = num_cands - len;
            NanMean *dst = strategy == CMI_RANK_UC ? total : per_user;
            double vals[N_MEAS];
            list_measures(ranked.data(), len, t, num_dropped, num_recs, vals);
            for (int m = 0; m < N_MEAS; ++m) dst[m].add(vals[m]);
            ++emitted;
        }
        // ucu: a test user contributes the NaN-skipping mean over its contexts -- NaN (skipped again) if none of
        // its contexts produced a list (Recommender.java:903-926)
        if (strategy == CMI_RANK_UCU && (q + 1 == nq || plan.qu[q + 1] != plan.qu[q])) flush_user();
    }
    for (int m = 0; m < N_MEAS; ++m) out[m] = total[m].value();
    (void)emitted;
}

template <typename T>
hipError_t rank_run_device(hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, const RankPlan &plan, const RankOperands<T> &ops,
                           double thold, int topn, std::vector<int32_t> &top_idx, std::vector<double> &top_score,
                           std::vector<int32_t> &top_count, float *ms, double *flops) {
    const std::vector<int32_t> &cand = plan.cand, &qu = plan.qu, &qc = plan.qc, &excl_idx = plan.excl_idx;
    const std::vector<int64_t> &excl_ptr = plan.excl_ptr;
    const int nc = (int)cand.size();
    const int64_t nq = (int64_t)qu.size();
    // operand rows are zero-padded to the GEMM's k step; row counts rounded up to the 128-row block tile (pad rows are
    // never written back)
    const int kp = (ops.k_logical + 15) / 16 * 16; // multiple of both kernels' k step
    auto up128 = [](int64_t v) { return (size_t)((v + 127) / 128 * 128); };
    // fp32 state with many candidates: the slab-free form (rank_kernels.hip, RankFilter) -- a sample of the candidates gives every query a
    // lower bound of its N-th best score, the full contraction then appends only the scores that can still make the list; the
    // [queries x candidates] score slab (21.5 GB for 270 K queries x 20 K items) is never written or re-read
    bool filtered = false;
    if constexpr (sizeof(T) == 4) filtered = rank_filter_usable(nc, kp, topn);
    const int ns = filtered ? rank_filter_sample(nc) : nc;
    const int cap = 1024;
    // query batch: keep the score slab (or the sample slab + the survivor lists) around 1 GiB
    const int64_t per_query = filtered ? (int64_t)ns * 4 + (int64_t)cap * 8 : (int64_t)nc * (int64_t)sizeof(T);
    int64_t bq = std::max<int64_t>(64, ((int64_t)1 << 30) / per_query);
    bq = std::min<int64_t>(bq, nq);
    if (const char *e = getenv("CMI_RANK_BATCH")) bq = std::max<int64_t>(1, std::min<int64_t>(atoll(e), nq));

    T *dA = nullptr, *dB = nullptr, *dS = nullptr, *drc = nullptr;
    int32_t *dcand = nullptr, *dqu = nullptr, *dqc = nullptr, *dexcl = nullptr, *dtop = nullptr, *dcount = nullptr;
    int64_t *dexptr = nullptr;
    double *dscore = nullptr;
    float *dtau = nullptr;
    int *dcnt = nullptr, *dover = nullptr;
    int2 *dlist = nullptr;
    T *dS_full = nullptr; // slab form, allocated only if a filtered batch overflows
    hipError_t e = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, std::max<size_t>(bytes, 8));
    };
    alloc((void **)&dB, up128(nc) * kp * sizeof(T));
    alloc((void **)&dA, up128(bq) * kp * sizeof(T));
    alloc((void **)&dS, (size_t)bq * (size_t)ns * sizeof(T));
    if (filtered) {
        alloc((void **)&dtau, (size_t)bq * 4);
        alloc((void **)&dcnt, (size_t)bq * 4);
        alloc((void **)&dover, 4);
        alloc((void **)&dlist, (size_t)bq * cap * sizeof(int2));
    }
    alloc((void **)&drc, (size_t)bq * sizeof(T));
    alloc((void **)&dcand, (size_t)nc * 4);
    alloc((void **)&dqu, (size_t)nq * 4);
    alloc((void **)&dqc, (size_t)nq * 4);
    alloc((void **)&dexptr, (size_t)(nq + 1) * 8);
    alloc((void **)&dexcl, excl_idx.size() * 4);
    alloc((void **)&dtop, (size_t)nq * topn * 4);
    alloc((void **)&dscore, (size_t)nq * topn * 8);
    alloc((void **)&dcount, (size_t)nq * 4);
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
    if (e == hipSuccess) e = hipEventRecord(ev0, stream);
    for (int64_t q0 = 0; q0 < nq && e == hipSuccess; q0 += bq) {
        const int n = (int)std::min<int64_t>(bq, nq - q0);
        e = ops.build_queries(dA, drc, dqu + q0, dqc + q0, n, kp, stream);
        if (e != hipSuccess) break;
        if (filtered) {
            if constexpr (sizeof(T) == 4) {
                e = hipMemsetAsync(dover, 0, 4, stream);
                if (e == hipSuccess)
                    e = rank_launch_score_filtered((const float *)dA, (const float *)dB, (const float *)drc, (float *)dS, n, nc, kp, dexptr, dexcl,
                                                   (int)q0, thold, topn, dtau, dcnt, dlist, cap, dover, dtop, dscore, dcount, stream);
                int over = 0;
                if (e == hipSuccess) e = hipMemcpyAsync(&over, dover, 4, hipMemcpyDeviceToHost, stream);
                if (e == hipSuccess) e = hipStreamSynchronize(stream); // one host round trip per ~100 K queries
                if (e == hipSuccess && over > 0) {
                    // some query keeps more than `cap` candidates at or above its bound (heavy ties, or too few qualifying scores in the
                    // sample): this batch goes through the slab form, in sub-batches of the slab's size
                    const int64_t sub = std::max<int64_t>(64, ((int64_t)1 << 30) / ((int64_t)nc * 4));
                    if (!dS_full) e = hipMalloc((void **)&dS_full, (size_t)std::min<int64_t>(sub, bq) * (size_t)nc * 4);
                    for (int64_t s0 = 0; s0 < n && e == hipSuccess; s0 += sub) {
                        const int m = (int)std::min<int64_t>(sub, n - s0);
                        e = rank_launch_score<T>(dA + (size_t)s0 * kp, dB, drc + s0, dS_full, m, nc, kp, dexptr, dexcl, (int)(q0 + s0), thold, topn,
                                                 dtop, dscore, dcount, stream);
                    }
                }
            }
        } else {
            e = rank_launch_score<T>(dA, dB, drc, dS, n, nc, kp, dexptr, dexcl, (int)q0, thold, topn, dtop, dscore, dcount, stream);
        }
    }
    if (e == hipSuccess) e = hipEventRecord(ev1, stream);
    top_idx.resize((size_t)nq * topn);
    top_score.resize((size_t)nq * topn);
    top_count.resize((size_t)nq);
    if (e == hipSuccess && nq) e = hipMemcpyAsync(top_idx.data(), dtop, (size_t)nq * topn * 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && nq) e = hipMemcpyAsync(top_score.data(), dscore, (size_t)nq * topn * 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && nq) e = hipMemcpyAsync(top_count.data(), dcount, (size_t)nq * 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e == hipSuccess && ms) e = hipEventElapsedTime(ms, ev0, ev1);
    if (flops) *flops = 2.0 * (double)nq * (double)nc * (double)kp;
    void *ptrs[] = {dA, dB, dS, drc, dcand, dqu, dqc, dexptr, dexcl, dtop, dscore, dcount, dtau, dcnt, dover, dlist, dS_full};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    return e;
}
template hipError_t rank_run_device<float>(hipStream_t, hipEvent_t, hipEvent_t, const RankPlan &, const RankOperands<float> &, double, int,
                                           std::vector<int32_t> &, std::vector<double> &, std::vector<int32_t> &, float *, double *);
template hipError_t rank_run_device<double>(hipStream_t, hipEvent_t, hipEvent_t, const RankPlan &, const RankOperands<double> &, double, int,
                                            std::vector<int32_t> &, std::vector<double> &, std::vector<int32_t> &, float *, double *);


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

extern "C" int cmi_eval_rankings(cmi_handle h, int64_t n_train, const int32_t *tu, const int32_t *tj,
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
        for (int64_t t = 0; t < n; ++t) {
            if (u[t] < 0 || u[t] >= h->n_users || j[t] < 0 || j[t] >= h->n_items)
                CMI_FAIL(h, CMI_E_INVALID, "eval_rankings: %s user/item id out of range at tuple %lld", what, (long long)t);
            if (c[t] < 0 || (contextual && c[t] >= h->n_ctx))
                CMI_FAIL(h, CMI_E_INVALID, "eval_rankings: %s context id %d out of range at tuple %lld", what, c[t], (long long)t);
        }
        return CMI_OK;
    };
    if (int rc = check(n_train, tu, tj, tctx, "train")) return rc;
    if (int rc = check(n_test, su, sj, sctx, "test")) return rc;
    CMI_HIP(h, hipSetDevice(h->device));
    if (n_queries) *n_queries = 0;

    RankPlan plan;
    rank_build_plan(h->n_users, h->n_items, RankTuples{n_train, tu, tj, tctx, tr}, RankTuples{n_test, su, sj, sctx, sr}, bin_thold,
                    num_ignore, plan);
    std::vector<int32_t> top_idx, top_count;
    std::vector<double> top_score;
    if (!plan.qu.empty() && !plan.cand.empty()) {
        const bool ic_used = h->state[CMI_STATE_IC_BIAS] != nullptr;
        const int k_logical = ext ? h->k + (h->model == CMI_MODEL_SVDPP ? 1 : 0)
                                  : h->k + 1 + (ic_used ? h->n_conds : 0); // [factors | 1 or itemBias | one-hot conditions or icBias row]
        hipError_t e;
        if (ext && h->f64) e = rank_run_device<double>(h->stream, h->ev0, h->ev1, plan, ext_operands<double>(h, k_logical), bin_thold, num_recs,
                                                       top_idx, top_score, top_count, &h->last_rank_ms, &h->last_rank_flops);
        else if (ext) e = rank_run_device<float>(h->stream, h->ev0, h->ev1, plan, ext_operands<float>(h, k_logical), bin_thold, num_recs, top_idx,
                                                 top_score, top_count, &h->last_rank_ms, &h->last_rank_flops);
        else if (h->f64) e = rank_run_device<double>(h->stream, h->ev0, h->ev1, plan, mf_operands<double>(h, k_logical, contextual, ic_used), bin_thold,
                                                num_recs, top_idx, top_score, top_count, &h->last_rank_ms, &h->last_rank_flops);
        else e = rank_run_device<float>(h->stream, h->ev0, h->ev1, plan, mf_operands<float>(h, k_logical, contextual, ic_used), bin_thold,
                                        num_recs, top_idx, top_score, top_count, &h->last_rank_ms, &h->last_rank_flops);
        CMI_HIP(h, e);
    } else {
        top_count.assign(plan.qu.size(), 0);
    }
    rank_metrics(plan, strategy, num_recs, top_idx, top_score, top_count, out, q_user, q_ctx, q_count, top_items, top_scores);
    if (n_queries) *n_queries = (int64_t)plan.qu.size();
    return CMI_OK;
}

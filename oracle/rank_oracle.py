"""CPU restatement of the reference's top-N ranking evaluation.  TEST INFRASTRUCTURE ONLY, PARITY UNPINNED by the
reference (no tests, no JVM); the metric definitions were read from the bytecode of the vendored
lib/happy.coding.utils-1.2.6.jar (happy.coding.math.Measures / Stats, happy.coding.io.Lists) and from
src/carskit/eval/Measures.java; hand-computed metric values pin them in tests/test_ranking.py.

  eval_rankings   Recommender.evalRankings            src/carskit/generic/Recommender.java:668-964
  *_at wrappers   carskit.eval.Measures               src/carskit/eval/Measures.java:13-67 (truncate to the top n first)
  hits_at/prec_at/recall_at/auc/ap/ndcg/rr            happy/coding/math/Measures.class (PrecAt, RecallAt, HitsAt, AUC, AP,
                                                      nDCG, IDCG, RR), Maths.log(x,2) = Math.log(x)/Math.log(2)
  mean            happy.coding.math.Stats.mean(Collection): NaN entries are SKIPPED, result = sum / #non-NaN (0/0 = NaN)
  candidate order java.util.HashSet<Integer> iteration order (hash = value, spread h ^ h>>>16, table doubled at load 0.75)
"""
import math


def hits_at(ranked, truth, n):
    hits = 0
    ts = set(truth)
    for i, item in enumerate(ranked):
        if item in ts:
            if i >= n:
                break
            hits += 1
    return hits


def prec_at(ranked, truth, n):
    return hits_at(ranked, truth, n) / (n + 0.0)


def recall_at(ranked, truth, n):
    return hits_at(ranked, truth, n) / (len(truth) + 0.0)


def top_n(ranked, n):
    return ranked[:min(n, len(ranked))]


def auc(ranked, truth, num_dropped):
    ts, rs = set(truth), set(ranked)
    num_rele = sum(1 for t in truth if t in rs)                 # Lists.overlapSize(groundTruth, rankedList)
    num_eval_items = len(ranked) + num_dropped
    num_eval_pairs = (num_eval_items - num_rele) * num_rele
    if num_eval_pairs < 0:
        raise ValueError("num_eval_pairs cannot be less than 0")
    if num_eval_pairs == 0:
        return 0.5
    correct = hits = 0
    for item in ranked:
        if item not in ts:
            correct += hits
        else:
            hits += 1
    num_miss = sum(1 for t in truth if t not in rs)             # Lists.exceptSize(groundTruth, rankedList)
    correct += hits * (num_dropped - num_miss)
    return (correct + 0.0) / num_eval_pairs


def ap(ranked, truth):
    ts = set(truth)
    hits, s = 0, 0.0
    for i, item in enumerate(ranked):
        if item in ts:
            hits += 1
            s += hits / (i + 1.0)
    return s / len(truth) if hits > 0 else 0.0


def _log2(x):
    return math.log(x) / math.log(2)


def idcg(n):
    s = 0.0
    for i in range(n):
        s += 1 / _log2(i + 2)
    return s


def ndcg(ranked, truth):
    ts = set(truth)
    dcg = 0.0
    for i, item in enumerate(ranked):
        if item in ts:
            dcg += 1 / _log2(i + 2)
    return dcg / idcg(len(truth))


def rr(ranked, truth):
    ts = set(truth)
    for i, item in enumerate(ranked):
        if item in ts:
            return 1 / (i + 1.0)
    return 0.0


def mean(xs):
    s, c = 0.0, 0
    for x in xs:
        if not math.isnan(x):
            s += x
            c += 1
    return s / c if c else float("nan")


def java_int_hashset_order(values):
    """Iteration order of a java.util.HashSet<Integer> after add()ing `values` in order (duplicates ignored)."""
    seen, keys = set(), []
    for v in values:
        if v not in seen:
            seen.add(v)
            keys.append(v)
    cap = 16
    while len(keys) > 0.75 * cap:
        cap <<= 1
    def bucket(v):
        h = v & 0xFFFFFFFF
        return (h ^ (h >> 16)) & (cap - 1)
    return [k for _, _, k in sorted((bucket(k), i, k) for i, k in enumerate(keys))]


MEASURES = ("Pre5", "Pre10", "PreN", "Rec5", "Rec10", "RecN", "AUC5", "AUC10", "AUCN", "MAP5", "MAP10", "MAPN",
            "NDCG5", "NDCG10", "NDCGN", "MRR5", "MRR10", "MRRN")


def eval_rankings(predict, train, test, bin_thold=-1.0, num_recs=10, strategy="ucu", num_ignore=0):
    """predict(u, j, c) -> unbounded score (ranking() = predict(u,j,c,false), Recommender.java:1016-1018);
    train/test: iterables of (u, j, ctx, r) in CRS order.  Returns {measure: value} and the per-query top lists.
    num_recs must be >= 1: with -topN <= 0 the reference's cut-off list holds a non-positive n and
    carskit.eval.Measures.getTopNList (Measures.java:13-16) throws (n<0) or degenerates (n=0)."""
    if num_recs < 1:
        raise ValueError("-topN must be >= 1")
    train = [t for t in train if t[3] != 0.0]
    test = [t for t in test if t[3] != 0.0]
    uci, order_u = {}, []                      # test positives: user -> ctx -> items (rate > threshold)
    for (u, j, c, r) in test:
        if r > bin_thold:
            if u not in uci:
                uci[u] = {}
                order_u.append(u)
            uci[u].setdefault(c, []).append(j)
    uci_train = {}
    for (u, j, c, r) in train:
        uci_train.setdefault(u, {}).setdefault(c, set()).add(j)
    cand = java_int_hashset_order([j for (_, j, _, _) in train])    # rateDao.getItemList(trainMatrix)
    if num_ignore > 0:                                             # drop the most popular items (Recommender.java:720-735)
        deg = {}
        for (_, j, _, _) in train:
            deg[j] = deg.get(j, 0) + 1
        by_deg = sorted(cand, key=lambda j: -deg[j])                # stable, descending by training degree
        dropped = set(by_deg[:num_ignore])
        cand = [j for j in cand if j not in dropped]
    cand_set = set(cand)
    lists = {m: [] for m in MEASURES}
    tops = {}
    # uciList is a HashMap<Integer, HashMultimap<Integer, Integer>>: users, and a user's contexts, are visited in java.util.HashMap order
    # (Recommender.java:738, 770) -- which fixes the summation order of the means below
    for u in java_int_hashset_order(order_u):
        c_lists = {m: [] for m in MEASURES}
        for c in java_int_hashset_order(list(uci[u])):
            pos_items = uci[u][c]
            num_cands = len(cand)
            correct = [j for j in pos_items if j in cand_set]
            if not correct:
                continue
            rated = uci_train.get(u, {}).get(c, set())
            scores = []
            for j in cand:
                if j not in rated:
                    s = predict(u, j, c)
                    if not math.isnan(s) and s > bin_thold:
                        scores.append((j, s))
                else:
                    num_cands -= 1
            if not scores:
                continue
            scores.sort(key=lambda kv: -kv[1])                   # Collections.sort, stable, descending by value
            recomd = scores if (num_recs <= 0 or len(scores) <= num_recs) else scores[:num_recs]
            ranked = [j for j, _ in recomd]
            tops[(u, c)] = recomd
            num_dropped = num_cands - len(ranked)
            vals = {}
            for tag, n in (("5", 5), ("10", 10), ("N", num_recs)):
                vals["Pre" + tag] = prec_at(ranked, correct, n)
                vals["Rec" + tag] = recall_at(ranked, correct, n)
                vals["AUC" + tag] = auc(top_n(ranked, n), correct, num_dropped)
                vals["MAP" + tag] = ap(top_n(ranked, n), correct)
                vals["NDCG" + tag] = ndcg(top_n(ranked, n), correct)
                vals["MRR" + tag] = rr(top_n(ranked, n), correct)
            target = lists if strategy == "uc" else c_lists
            for m in MEASURES:
                target[m].append(vals[m])
        if strategy != "uc":
            for m in MEASURES:
                lists[m].append(mean(c_lists[m]))
    out = {m: mean(lists[m]) for m in MEASURES}
    out.update({"D5": 0.0, "D10": 0.0, "DN": 0.0})
    return out, tops

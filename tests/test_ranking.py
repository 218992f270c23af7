"""CPU tests of the ranking-evaluation oracle (oracle/rank_oracle.py): metric formulas against hand-computed values,
the HashSet<Integer> candidate order, and the evalRankings control flow (Recommender.java:668-964) on a tiny case
worked out by hand.  The product's HashSet-order entry point (pure host code) is checked against the oracle here too."""
import math

import numpy as np
import pytest

from carskit_amd import capi
from oracle import rank_oracle as ro

RANKED = [3, 1, 4, 2, 5]
TRUTH = [1, 5, 9]


def test_hits_prec_recall_hand_values():
    assert ro.hits_at(RANKED, TRUTH, 5) == 2
    assert ro.hits_at(RANKED, TRUTH, 3) == 1          # item 5 sits at index 4 >= 3 -> loop breaks
    assert ro.hits_at(RANKED, TRUTH, 10) == 2
    assert ro.prec_at(RANKED, TRUTH, 5) == pytest.approx(0.4, abs=0)
    assert ro.prec_at(RANKED, TRUTH, 10) == pytest.approx(0.2, abs=0)   # denominator is n, not the list length
    assert ro.recall_at(RANKED, TRUTH, 5) == pytest.approx(2 / 3, abs=1e-16)


def test_ap_ndcg_rr_hand_values():
    assert ro.ap(RANKED, TRUTH) == pytest.approx((1 / 2 + 2 / 5) / 3, abs=1e-16)
    dcg = 1 / (math.log(3) / math.log(2)) + 1 / (math.log(6) / math.log(2))
    idcg = 1 / 1.0 + 1 / (math.log(3) / math.log(2)) + 1 / 2.0
    assert ro.ndcg(RANKED, TRUTH) == pytest.approx(dcg / idcg, abs=1e-15)
    assert ro.ndcg(RANKED, TRUTH) == pytest.approx(0.47762, abs=1e-5)
    assert ro.rr(RANKED, TRUTH) == 0.5
    assert ro.rr([7, 8], TRUTH) == 0.0
    assert ro.ap([7, 8], TRUTH) == 0.0


def test_auc_hand_values():
    # 2 relevant in the list, 13 irrelevant overall (3 listed + 10 dropped), item 9 missing: 26 pairs, 20 correct
    assert ro.auc(RANKED, TRUTH, 10) == pytest.approx(20 / 26, abs=1e-16)
    assert ro.auc([7, 8], TRUTH, 10) == 0.5          # no relevant item in the list -> zero pairs -> 0.5
    assert ro.auc([1, 5], [1, 5], 0) == 0.5          # everything relevant -> zero pairs
    assert ro.auc([1, 7], [1], 0) == 1.0
    assert ro.auc([7, 1], [1], 0) == 0.0


def test_mean_skips_nan():
    assert ro.mean([1.0, float("nan"), 3.0]) == 2.0
    assert math.isnan(ro.mean([]))
    assert math.isnan(ro.mean([float("nan")]))


def test_hashset_order_small_tables():
    # 5 keys -> table of 16: bucket = key & 15, so 33 (bucket 1) precedes 2, and 18 shares bucket 2 after 2
    assert ro.java_int_hashset_order([2, 33, 18, 7, 2]) == [33, 2, 18, 7]
    # 13 keys -> resized to 32 (13 > 12)
    keys = [40, 8, 72, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11]
    assert ro.java_int_hashset_order(keys) == [1, 2, 3, 4, 5, 6, 7, 40, 8, 72, 9, 10, 11]
    # high bits are folded in: 65536 -> h ^ h>>>16 = 65537 -> bucket 1
    assert ro.java_int_hashset_order([65536, 0, 1]) == [0, 65536, 1]


def test_hashset_order_library_matches_oracle():
    rng = np.random.default_rng(5)
    for n, hi in ((0, 10), (1, 10), (12, 100), (13, 100), (200, 1000), (5000, 200000)):
        v = rng.integers(0, hi, size=n)
        assert capi.java_int_hashset_order(v).tolist() == ro.java_int_hashset_order(v.tolist())


def _tiny():
    # items 0..5; user 0 in ctx 0 rated {0,1} in training; test positives: (0, ctx0): items 2 and 4; (0, ctx1): item 3;
    # (1, ctx0): item 5 rated 1.0 (below threshold 2.5) -> no query for user 1
    train = [(0, 0, 0, 4.0), (0, 1, 0, 3.0), (1, 2, 0, 5.0), (1, 3, 1, 2.0), (1, 4, 0, 4.0), (1, 5, 1, 4.0)]
    test = [(0, 2, 0, 5.0), (0, 4, 0, 3.0), (0, 3, 1, 4.0), (1, 5, 0, 1.0)]
    score = {0: 9.0, 1: 8.0, 2: 3.0, 3: 5.0, 4: 4.0, 5: 6.0}
    return train, test, (lambda u, j, c: score[j] + (0.5 if c == 1 and j == 2 else 0.0))


def test_eval_rankings_tiny_by_hand():
    train, test, predict = _tiny()
    m, tops = ro.eval_rankings(predict, train, test, bin_thold=2.5, num_recs=3, strategy="uc")
    # query (0, ctx0): rated {0,1} are skipped -> 4 candidates 5,3,4,2 by score 6,5,4,3; top-3 = [5,3,4]; truth {2,4}
    assert [j for j, _ in tops[(0, 0)]] == [5, 3, 4]
    # query (0, ctx1): nothing rated in ctx1 -> candidates 0,1,5,3,4,2 (2 scores 3.5): top-3 = [0,1,5]; truth {3}
    assert [j for j, _ in tops[(0, 1)]] == [0, 1, 5]
    assert (1, 0) not in tops
    # Pre@3: 1/3 and 0; Pre@5 uses the same 3-long list with denominator 5
    assert m["PreN"] == pytest.approx((1 / 3 + 0) / 2, abs=1e-16)
    assert m["Pre5"] == pytest.approx((1 / 5 + 0) / 2, abs=1e-16)
    assert m["RecN"] == pytest.approx((1 / 2 + 0) / 2, abs=1e-16)
    assert m["MRRN"] == pytest.approx((1 / 3 + 0) / 2, abs=1e-16)
    assert m["MAPN"] == pytest.approx(((1 / 3) / 2 + 0) / 2, abs=1e-16)
    # AUC (0,ctx0): 4 evaluated items, 1 relevant listed, 1 dropped (item 2, relevant & missing): pairs (4-1)*1 = 3,
    # correct = 0 (hit comes last) + 1*(1 dropped - 1 missed) = 0 -> 0;  (0,ctx1): no relevant listed -> 0.5
    assert m["AUCN"] == pytest.approx((0.0 + 0.5) / 2, abs=1e-16)
    # ucu: both queries belong to user 0 -> same numbers here
    m2, _ = ro.eval_rankings(predict, train, test, bin_thold=2.5, num_recs=3, strategy="ucu")
    assert m2["PreN"] == m["PreN"] and m2["AUCN"] == m["AUCN"]


def test_eval_rankings_threshold_filters_scores():
    train, test, predict = _tiny()
    # scores <= 4.5 are not recommended: (0,ctx0) keeps 5 and 3 only
    _, tops = ro.eval_rankings(predict, train, test, bin_thold=4.5, num_recs=3, strategy="uc")
    assert [j for j, _ in tops[(0, 0)]] == [5, 3]


def test_eval_rankings_ignore_popular_items():
    train, test, predict = _tiny()
    train = train + [(2, 5, 0, 3.0), (2, 5, 1, 3.0)]     # item 5 becomes the most rated
    _, tops = ro.eval_rankings(predict, train, test, bin_thold=2.5, num_recs=3, strategy="uc", num_ignore=1)
    assert [j for j, _ in tops[(0, 0)]] == [3, 4, 2]
    assert [j for j, _ in tops[(0, 1)]] == [0, 1, 3]


def test_eval_rankings_rejects_nonpositive_topn():
    train, test, predict = _tiny()
    with pytest.raises(ValueError):
        ro.eval_rankings(predict, train, test, num_recs=0)


@pytest.mark.parametrize("seed,num_ignore,thold", [(1, 0, 2.5), (2, 3, 3.5), (3, 0, -1.0), (4, 1, 4.5)])
def test_rank_plan_bookkeeping_matches_a_python_derivation(seed, num_ignore, thold):
    """cmi_rank_plan (host-only part of cmi_eval_rankings): candidates, queries, correct items and per-query exclusions
    against a direct Python derivation from the rules of Recommender.evalRankings (Recommender.java:704-816)."""
    rng = np.random.default_rng(seed)
    n_users, n_items, n = 40, 500, 900
    items = rng.choice(n_items, size=60, replace=False)          # sparse ids: HashSet order != ascending
    u = rng.integers(0, n_users, n).astype(np.int32)
    j = items[rng.integers(0, len(items), n)].astype(np.int32)
    c = rng.integers(0, 6, n).astype(np.int32)
    r = rng.integers(1, 6, n).astype(np.float64)
    keep = np.unique(np.stack([u, j, c], 1), axis=0, return_index=True)[1]
    keep.sort()
    u, j, c, r = u[keep], j[keep], c[keep], r[keep]
    tr = rng.random(len(u)) < 0.7
    train, test = (u[tr], j[tr], c[tr], r[tr]), (u[~tr], j[~tr], c[~tr], r[~tr])
    cand, queries = capi.rank_plan(n_users, n_items, train, test, thold, num_ignore)
    # expected candidates
    exp_cand = ro.java_int_hashset_order(train[1].tolist())
    if num_ignore:
        deg = {}
        for jj in train[1].tolist():
            deg[jj] = deg.get(jj, 0) + 1
        drop = set(sorted(exp_cand, key=lambda x: -deg[x])[:num_ignore])
        exp_cand = [x for x in exp_cand if x not in drop]
    assert cand == exp_cand and cand != sorted(cand)
    pos = {x: i for i, x in enumerate(cand)}
    # expected queries
    truth, rated = {}, {}
    for uu, jj, cc, rr in zip(*(a.tolist() for a in test)):
        if rr > thold and jj in pos:
            truth.setdefault((uu, cc), set()).add(jj)
    for uu, jj, cc, rr in zip(*(a.tolist() for a in train)):
        if jj in pos:
            rated.setdefault((uu, cc), set()).add(pos[jj])
    got = {(qu, qc): (t, e) for qu, qc, t, e in queries}
    assert set(got) == set(truth) and len(queries) == len(truth)
    assert [(q[0], q[1]) for q in queries] == sorted(truth)       # (user, context) order
    for key, items_ in truth.items():
        t, e = got[key]
        assert t == sorted(items_) and sorted(e) == sorted(rated.get(key, set())) and len(set(e)) == len(e)
    # the plan is built in ranges of users on the host's cores (forced here: the input is far below the size that would use them):
    # element for element the serial result, whatever the number of ranges -- more ranges than users included
    import os
    for nt, deg_ranges in (("3", None), ("7", None), ("64", None), ("7", "2"), ("5", "1")):
        os.environ["CMI_HOST_THREADS"] = nt
        if deg_ranges:      # the form for huge catalogues: fewer, larger ranges record the item degrees in a pass of their own
            os.environ["CMI_PLAN_DEG_RANGES"] = deg_ranges
        try:
            assert capi.rank_plan(n_users, n_items, train, test, thold, num_ignore) == (cand, queries)
        finally:
            del os.environ["CMI_HOST_THREADS"]
            os.environ.pop("CMI_PLAN_DEG_RANGES", None)


def test_list_measures_library_matches_oracle_formulas():
    """The product's per-list metric code (host C++) against the oracle's formulas on random lists."""
    rng = np.random.default_rng(9)
    for _ in range(300):
        num_recs = int(rng.choice([1, 3, 5, 7, 10, 25]))
        universe = rng.permutation(60)
        ranked = universe[:int(rng.integers(0, num_recs + 1))].tolist()
        truth = rng.choice(60, size=int(rng.integers(1, 8)), replace=False).tolist()
        dropped = int(rng.integers(0, 50))
        got = capi.rank_list_measures(ranked, truth, dropped, num_recs)
        for tag, n in (("5", 5), ("10", 10), ("N", num_recs)):
            top = ro.top_n(ranked, n)
            want = {"Pre": ro.prec_at(ranked, truth, n), "Rec": ro.recall_at(ranked, truth, n), "AUC": ro.auc(top, truth, dropped),
                    "MAP": ro.ap(top, truth), "NDCG": ro.ndcg(top, truth), "MRR": ro.rr(top, truth)}
            for m, v in want.items():
                assert got[m + tag] == pytest.approx(v, abs=1e-15), (m + tag, ranked, truth, dropped)
    # the hand-computed list of the oracle tests, through the library
    g = capi.rank_list_measures([3, 1, 4, 2, 5], [1, 5, 9], 10, 5)
    assert g["Pre5"] == 0.4 and g["MRR5"] == 0.5 and g["AUC5"] == pytest.approx(20 / 26, abs=1e-16)
    assert g["MAP5"] == pytest.approx((1 / 2 + 2 / 5) / 3, abs=1e-16)


def test_rank_plan_is_the_same_for_any_number_of_host_threads_on_a_larger_input(monkeypatch):
    """The plan's threaded form (tuple ranges -> user buckets -> queries; the candidates' first-seen order merged over ranges) on an input
    large enough to fill every range and bucket: 6 000 users (more than 256 buckets' worth), sparse item ids (HashSet order != ascending),
    duplicate (user, context, item) cells, zero ratings in the training set, -ignore: 1, 2, 7 and 16 ranges give the same arrays."""
    rng = np.random.default_rng(12)
    n_users, n_items, n = 6000, 5000, 120_000
    items = rng.choice(n_items, size=1200, replace=False)
    u = rng.integers(0, n_users, n).astype(np.int32)
    j = items[rng.integers(0, len(items), n)].astype(np.int32)
    c = rng.integers(0, 40, n).astype(np.int32)
    r = rng.integers(0, 6, n).astype(np.float64)          # zeros included: a sparse matrix holds no zero entries
    cut = int(0.8 * n)
    train, test = (u[:cut], j[:cut], c[:cut], r[:cut]), (u[cut:], j[cut:], c[cut:], r[cut:])
    ref = None
    for nt in ("1", "2", "7", "16"):
        monkeypatch.setenv("CMI_HOST_THREADS", nt)
        for ignore in (0, 25):
            got = capi.rank_plan(n_users, n_items, train, test, 2.5, ignore)
            if ref is None or ignore not in ref:
                ref = ref or {}
                ref[ignore] = got
                assert len(got[1]) > 5000
            else:
                assert got == ref[ignore], (nt, ignore)

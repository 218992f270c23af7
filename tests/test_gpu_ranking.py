"""GPU parity of cmi_eval_rankings (Recommender.evalRankings, Recommender.java:668-964) against oracle/rank_oracle.py
driven by the C oracle's scalar predict().

Bars: fp64 state -> identical top-N item lists, scores within 1e-12, all 18 measures within 1e-12 (the GPU sums the
predict() terms in a different order: dot product tree/tiled, biases folded into the contraction);
fp32 state -> scores within 1e-4, measures within 0.02 (near-tied neighbours may swap)."""
import math
import os

import numpy as np
import pytest

from carskit_amd import capi, synth
from oracle import oracle_c, rank_oracle
from tests import util

pytestmark = pytest.mark.gpu
F64, SERIAL, STRICT = capi.FLAG_STATE_F64, capi.FLAG_SCHED_SERIAL, capi.FLAG_STRICT


def _setup(model, k, flags, epochs=2, n_users=60, n_items=90, n=2500, seed=3):
    data = synth.generate(n_users, n_items, 3, 3, n, seed=seed)
    train, test = synth.split(data, 0.25)
    state = synth.init_state(model, train, k, seed=9)
    gm = 0.0 if model == "PMF" else oracle_c.global_mean(train.r)
    orc = util.c_oracle(model, train, k, state, gm)
    u, j, ctx, r = util.tuples_for(model, train)
    if model == "CAMF_C":
        flags |= SERIAL
    inst = capi.Instance(model, k, train.n_users, train.n_items, train.n_conds, flags=flags)
    inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
    if model in util.TWO_D:
        inst.set_ratings(u, j, None, r)
    else:
        inst.set_ratings(u, j, ctx, r, train.ctx_ptr, train.ctx_conds)
    inst.set_states(state)
    for _ in range(epochs):
        orc.epoch(util.LR)
        inst.train_epoch(util.LR)
    return train, test, orc, inst


def _tuples(d):
    return list(zip(d.u.tolist(), d.j.tolist(), d.ctx.tolist(), d.r.tolist()))


def _arrays(d):
    return d.u, d.j, d.ctx, d.r


def _oracle_eval(orc, train, test, **kw):
    return rank_oracle.eval_rankings(lambda u, j, c: orc.predict(u, j, c), _tuples(train), _tuples(test), **kw)


def _assert_same(res, lists, ref, ref_lists, tol, score_tol, same_items=True):
    assert set(lists) == set(ref_lists)
    for key, ref_l in ref_lists.items():
        got = lists[key]
        assert len(got) == len(ref_l), key
        if same_items:
            assert [i for i, _ in got] == [i for i, _ in ref_l], key
        for (_, a), (_, b) in zip(got, ref_l):
            assert abs(a - b) <= score_tol, (key, a, b)
    for m in rank_oracle.MEASURES:
        a, b = res[m], ref[m]
        assert (math.isnan(a) and math.isnan(b)) or abs(a - b) <= tol, (m, a, b)
    assert res["D5"] == res["D10"] == res["DN"] == 0.0


@pytest.mark.parametrize("model", util.MODELS)
@pytest.mark.parametrize("strategy", ["ucu", "uc"])
def test_f64_rankings_match_oracle(model, strategy):
    train, test, orc, inst = _setup(model, 10, F64 | STRICT)
    thold = -1.0 if model == "PMF" else 2.5     # PMF scores are bare dot products, far below the rating scale
    ref, ref_lists = _oracle_eval(orc, train, test, bin_thold=thold, num_recs=10, strategy=strategy)
    res, lists = inst.eval_rankings(_arrays(train), _arrays(test), bin_thold=thold, num_recs=10, strategy=strategy,
                                    with_lists=True)
    assert len(ref_lists) > 20
    _assert_same(res, lists, ref, ref_lists, 1e-12, 1e-12)


@pytest.mark.parametrize("model", ["CAMF_CI", "CAMF_CUCI", "BiasedMF"])
@pytest.mark.parametrize("k,num_recs", [(3, 3), (64, 7), (70, 25), (10, 64), (10, 70)])
def test_f64_rankings_shapes_and_topn(model, k, num_recs):
    train, test, orc, inst = _setup(model, k, F64 | STRICT, epochs=1)
    ref, ref_lists = _oracle_eval(orc, train, test, bin_thold=-1.0, num_recs=num_recs)
    res, lists = inst.eval_rankings(_arrays(train), _arrays(test), bin_thold=-1.0, num_recs=num_recs, with_lists=True)
    _assert_same(res, lists, ref, ref_lists, 1e-12, 1e-12)


def test_query_batching_is_invisible():
    train, test, orc, inst = _setup("CAMF_CUCI", 10, F64 | STRICT)
    whole = inst.eval_rankings(_arrays(train), _arrays(test), bin_thold=2.5, with_lists=True)
    os.environ["CMI_RANK_BATCH"] = "7"
    try:
        split = inst.eval_rankings(_arrays(train), _arrays(test), bin_thold=2.5, with_lists=True)
        # ... and so are the host-side ranges: the plan in 5 ranges of users, every 7-query batch's measures in 5 ranges of queries,
        # computed behind the device's next batch
        os.environ["CMI_HOST_THREADS"] = "5"
        ranged = inst.eval_rankings(_arrays(train), _arrays(test), bin_thold=2.5, with_lists=True)
    finally:
        del os.environ["CMI_RANK_BATCH"]
        os.environ.pop("CMI_HOST_THREADS", None)
    assert whole == split == ranged


def test_threshold_and_ignore():
    train, test, orc, inst = _setup("CAMF_CI", 10, F64 | STRICT)
    for kw in (dict(bin_thold=3.5, num_recs=5), dict(bin_thold=2.5, num_recs=10, num_ignore=7),
               dict(bin_thold=4.2, num_recs=10, strategy="uc")):
        ref, ref_lists = _oracle_eval(orc, train, test, **kw)
        res, lists = inst.eval_rankings(_arrays(train), _arrays(test), with_lists=True, **kw)
        _assert_same(res, lists, ref, ref_lists, 1e-12, 1e-12)


def test_ties_follow_the_hashset_candidate_order():
    # all-zero model: every score equals the global mean, so the list is the first num_recs non-rated candidates
    # in HashSet<Integer> order; item ids are sparse (0..999) so that order is NOT ascending
    rng = np.random.default_rng(2)
    items = rng.choice(1000, size=40, replace=False)
    n = 400
    d = synth.generate(30, 40, 2, 3, n, seed=4)
    j = items[d.j].astype(np.int32)
    tr_mask = rng.random(len(j)) < 0.75
    train = (d.u[tr_mask], j[tr_mask], d.ctx[tr_mask], d.r[tr_mask])
    test = (d.u[~tr_mask], j[~tr_mask], d.ctx[~tr_mask], d.r[~tr_mask])
    inst = capi.Instance("BiasedMF", 4, 30, 1000, d.n_conds, flags=F64)
    inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, 3.0)
    inst.set_states({"P": np.zeros((30, 4)), "Q": np.zeros((1000, 4)), "userBias": np.zeros(30), "itemBias": np.zeros(1000)})
    tt = [list(zip(*(a.tolist() for a in x))) for x in (train, test)]
    ref, ref_lists = rank_oracle.eval_rankings(lambda u, jj, c: 3.0, tt[0], tt[1], bin_thold=2.5, num_recs=10)
    res, lists = inst.eval_rankings(train, test, bin_thold=2.5, num_recs=10, with_lists=True)
    order = rank_oracle.java_int_hashset_order(train[1].tolist())
    assert order != sorted(order)
    _assert_same(res, lists, ref, ref_lists, 1e-15, 0.0)


@pytest.mark.parametrize("model", ["CAMF_CI", "CAMF_CU", "BiasedMF"])
def test_f32_rankings_close_to_oracle(model):
    train, test, orc, inst = _setup(model, 32, 0, epochs=3)
    ref, ref_lists = _oracle_eval(orc, train, test, bin_thold=2.5, num_recs=10)
    res, lists = inst.eval_rankings(_arrays(train), _arrays(test), bin_thold=2.5, num_recs=10, with_lists=True)
    _assert_same(res, lists, ref, ref_lists, 0.02, 1e-4, same_items=False)
    same = sum([i for i, _ in lists[q]] == [i for i, _ in ref_lists[q]] for q in ref_lists)
    assert same >= 0.9 * len(ref_lists)


def test_degenerate_inputs_and_errors():
    train, test, orc, inst = _setup("CAMF_CI", 5, F64)
    # no positive test rating -> no query -> every measure is Stats.mean(empty) = NaN
    res = inst.eval_rankings(_arrays(train), _arrays(test), bin_thold=99.0)
    assert res["n_queries"] == 0 and all(math.isnan(res[m]) for m in rank_oracle.MEASURES)
    # a threshold above every score: queries exist but nothing is recommended
    res, lists = inst.eval_rankings(_arrays(train), (test.u, test.j, test.ctx, test.r + 50.0), bin_thold=40.0,
                                    with_lists=True)
    assert res["n_queries"] > 0 and not lists and math.isnan(res["Pre5"])
    # empty training set -> no candidates
    empty = tuple(a[:0] for a in _arrays(train))
    res = inst.eval_rankings(empty, _arrays(test), bin_thold=2.5)
    assert res["n_queries"] == 0
    with pytest.raises(capi.CmiError):
        inst.eval_rankings(_arrays(train), _arrays(test), num_recs=0)
    with pytest.raises(capi.CmiError):
        inst.eval_rankings((train.u + 10_000, train.j, train.ctx, train.r), _arrays(test))


@pytest.mark.parametrize("k,strategy", [(4, "ucu"), (33, "uc")])
def test_fm_rankings_match_oracle(k, strategy):
    """cmi_fm_eval_rankings: FM.predict (FM.java:93-113) as the scorer; the oracle ranks with the dense C restatement of
    FM.predict after the same sweeps.  fp64 on both sides: identical lists, scores and measures to 1e-9 (the FM model
    itself is held to 1e-8, tests/test_gpu_fm.py)."""
    from tests.test_oracle_fm import REGLF, REGLW, fm_init_model
    data = synth.generate(60, 90, 3, 3, 2500, seed=3)
    train, test = synth.split(data, 0.25)
    w0, w, V = fm_init_model(train.n_users, train.n_items, train.n_conds, k, 2)
    orc = oracle_c.FMOracle(k, train.n_users, train.n_items, train.n_conds, train.n_dims, train.u, train.j, train.ctx, train.r,
                            w0, w, V, REGLW, REGLF)
    g = capi.FMInstance(k, train.n_users, train.n_items, train.n_conds, train.n_dims)
    g.set_hparams(REGLW, REGLF)
    g.set_ratings(train.u, train.j, train.ctx, train.r)
    g.set_model(w0, w, V)
    orc.init()
    g.init()
    for _ in range(2):
        orc.sweep()
        g.sweep()
    ref, ref_lists = rank_oracle.eval_rankings(lambda u, j, c: orc.predict(u, j, c), _tuples(train), _tuples(test),
                                               bin_thold=-5.0, num_recs=10, strategy=strategy)
    res, lists = g.eval_rankings(_arrays(train), _arrays(test), bin_thold=-5.0, num_recs=10, strategy=strategy, with_lists=True)
    assert len(ref_lists) > 20    # (the reference's FM regularises with size*reg: its scores sit far below the rating scale)
    _assert_same(res, lists, ref, ref_lists, 1e-9, 1e-8)


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: v for k, v in env.items() if v is not None})
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_all_tied_scores_resolve_in_candidate_order():
    """An all-tied model: every candidate has the same score, so the lists are the first N candidates in HashSet order (the
    reference's stable descending sort over its candidate order)."""
    rng = np.random.default_rng(5)
    n_items = 1500
    d = synth.generate(25, n_items, 2, 3, 6000, seed=6)
    tr_mask = rng.random(d.n) < 0.8
    train = (d.u[tr_mask], d.j[tr_mask], d.ctx[tr_mask], d.r[tr_mask])
    test = (d.u[~tr_mask], d.j[~tr_mask], d.ctx[~tr_mask], d.r[~tr_mask])
    inst = capi.Instance("BiasedMF", 4, 25, n_items, d.n_conds)
    inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, 3.0)
    inst.set_states({"P": np.zeros((25, 4)), "Q": np.zeros((n_items, 4)), "userBias": np.zeros(25), "itemBias": np.zeros(n_items)})
    tt = [list(zip(*(a.tolist() for a in x))) for x in (train, test)]
    ref, ref_lists = rank_oracle.eval_rankings(lambda u, jj, c: 3.0, tt[0], tt[1], bin_thold=2.5, num_recs=10)
    res, lists = inst.eval_rankings(train, test, bin_thold=2.5, num_recs=10, with_lists=True)
    _assert_same(res, lists, ref, ref_lists, 1e-15, 0.0)
    # a second evaluation on the same handle reuses the cached workspace (smaller, then larger problem): same answer
    half = tuple(a[: len(a) // 2] for a in test)
    r2, l2 = inst.eval_rankings(train, half, bin_thold=2.5, num_recs=10, with_lists=True)
    ref2, ref_l2 = rank_oracle.eval_rankings(lambda u, jj, c: 3.0, tt[0], tt[1][: len(tt[1]) // 2], bin_thold=2.5, num_recs=10)
    _assert_same(r2, l2, ref2, ref_l2, 1e-15, 0.0)
    r3, l3 = inst.eval_rankings(train, test, bin_thold=2.5, num_recs=10, with_lists=True)
    _assert_same(r3, l3, ref, ref_lists, 1e-15, 0.0)
    hm = inst.last_rank_host_ms()
    assert hm["total"] >= hm["plan"] > 0.0 and hm["scoring_loop"] > 0.0




@pytest.mark.parametrize("model", ["CAMF_CI", "CAMF_CUCI", "CAMF_CU", "CAMF_C", "BiasedMF", "PMF"])
def test_split_form_agrees_with_the_per_query_contraction(model):
    """Round 4: for the MF family in fp32 the scores are contracted as S1[user] + S2[context] (rank_kernels.hip "the split form") instead
    of one dot product per query.  Same sum, another association: the lists of the two forms agree except at fp32 near-ties, scores to
    1e-5, measures to 0.01; exclusions (walked on the fly in the split form, masked in the slab in the other) give the same counts."""
    train, test, orc, inst = _setup(model, 32, 0, epochs=2, n_users=150, n_items=900, n=12000, seed=9)
    kw = dict(bin_thold=2.5, num_recs=10, with_lists=True)
    split = _with_env({"CMI_RANK_NO_SPLIT": None, "CMI_RANK_BATCH": "17"}, lambda: inst.eval_rankings(_arrays(train), _arrays(test), **kw))
    whole = _with_env({"CMI_RANK_NO_SPLIT": "1"}, lambda: inst.eval_rankings(_arrays(train), _arrays(test), **kw))
    assert split[0]["n_queries"] == whole[0]["n_queries"] > 0 and set(split[1]) == set(whole[1])
    same = 0
    for key, lst in whole[1].items():
        other = split[1][key]
        assert len(other) == len(lst)
        same += [i for i, _ in lst] == [i for i, _ in other]
        for (ia, sa), (ib, sb) in zip(lst, other):
            assert abs(sa - sb) <= 1e-5 * max(1.0, abs(sa))
    assert same >= 0.95 * len(whole[1])
    for m, v in whole[0].items():
        assert (math.isnan(v) and math.isnan(split[0][m])) or abs(split[0][m] - v) <= 0.01, m


@pytest.mark.parametrize("model,n_items", [("CAMF_CI", 5000), ("BiasedMF", 4200)])
def test_pruned_split_form_at_many_tiles_against_the_oracle(model, n_items):
    """The product path of the fp32 MF family -- split contraction on the matrix cores, tile-pruning selection over several 64-tile
    chunks, exclusions walked by the scalar cursor -- held to the ORACLE itself (rank_oracle over the C oracle's predict, fp64) at a
    candidate count the other oracle comparisons do not reach (they rank 90 items).  fp32 state against fp64: lists agree except at
    near-ties, scores to 1e-4, measures to 0.01."""
    train, test, orc, inst = _setup(model, 16, 0, epochs=2, n_users=40, n_items=n_items, n=9000, seed=21)
    kw = dict(bin_thold=2.5, num_recs=10)
    ref, ref_lists = _oracle_eval(orc, train, test, **kw)
    res, lists = inst.eval_rankings(_arrays(train), _arrays(test), with_lists=True, **kw)
    assert set(lists) == set(ref_lists) and len(ref_lists) > 20
    same = 0
    for key, ref_l in ref_lists.items():
        got = lists[key]
        assert len(got) == len(ref_l), key
        same += [i for i, _ in got] == [i for i, _ in ref_l]
        for (_, a), (_, b) in zip(got, ref_l):
            assert abs(a - b) <= 1e-4, (key, a, b)
    assert same >= 0.9 * len(ref_lists), (same, len(ref_lists))
    for m in rank_oracle.MEASURES:
        a, b = res[m], ref[m]
        assert (math.isnan(a) and math.isnan(b)) or abs(a - b) <= 0.01, (m, a, b)


def test_split_form_exclusions_with_sparse_item_ids():
    """The split form walks a query's exclusion list on the fly and needs it in ascending CANDIDATE POSITION; with sparse item ids the
    HashSet order of the candidates is not the order of the ids.  Users that rated many items in the context of their test query, an
    all-equal fp32 model (every score ties, so the list is the first num_recs non-excluded candidates: one missed exclusion shows): the
    split form, the per-query form and the oracle agree on every list."""
    rng = np.random.default_rng(21)
    n_users, id_space, n_items = 40, 5000, 300
    items = rng.choice(id_space, size=n_items, replace=False).astype(np.int32)
    d = synth.generate(n_users, n_items, 2, 2, 9000, seed=8)        # 4 contexts: many training items per (user, context)
    j = items[d.j]
    mask = rng.random(len(j)) < 0.85
    train = (d.u[mask], j[mask], d.ctx[mask], d.r[mask])
    test = (d.u[~mask], j[~mask], d.ctx[~mask], d.r[~mask])
    order = rank_oracle.java_int_hashset_order(train[1].tolist())
    assert order != sorted(order)
    inst = capi.Instance("CAMF_CI", 16, n_users, id_space, d.n_conds)            # fp32 state: the split form
    inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, 3.0)
    inst.set_ratings(train[0], train[1], train[2], train[3], d.ctx_ptr, d.ctx_conds)
    inst.set_states({"P": np.zeros((n_users, 16)), "Q": np.zeros((id_space, 16)), "userBias": np.zeros(n_users),
                     "icBias": np.zeros((id_space, d.n_conds))})
    tt = [list(zip(*(a.tolist() for a in x))) for x in (train, test)]
    ref, ref_lists = rank_oracle.eval_rankings(lambda u, jj, c: 3.0, tt[0], tt[1], bin_thold=2.5, num_recs=10)
    kw = dict(bin_thold=2.5, num_recs=10, with_lists=True)
    split = inst.eval_rankings(train, test, **kw)
    whole = _with_env({"CMI_RANK_NO_SPLIT": "1"}, lambda: inst.eval_rankings(train, test, **kw))
    plan = capi.rank_plan(n_users, id_space, train, test, 2.5, 0)
    assert max(len(e) for _, _, _, e in plan[1]) >= 20                            # long exclusion lists are in play
    assert any(e != sorted(e) for _, _, _, e in plan[1]) is False                 # ascending candidate positions
    for got in (split, whole):
        assert set(got[1]) == set(ref_lists)
        for key, lst in ref_lists.items():
            assert [i for i, _ in got[1][key]] == [i for i, _ in lst], key


@pytest.mark.parametrize("model,k,n_items,num_recs", [("CAMF_CI", 32, 900, 10), ("CAMF_CI", 128, 9000, 10), ("CAMF_CU", 16, 5000, 25),
                                                      ("BiasedMF", 8, 4500, 5), ("CAMF_CUCI", 32, 700, 64), ("PMF", 16, 130, 10)])
def test_tile_pruned_selection_equals_the_plain_selection(model, k, n_items, num_recs):
    """Round 6: the split form's selection skips tiles of 64 candidates whose bound (M1[user][tile] + M2[context][tile]) + c0 -- the row
    maxima the contractions' epilogues write, combined in the score's own floating-point operations -- cannot beat the query's current
    N-th best (rank_topn_split_pruned).  A bound, not an approximation: lists, scores, counts and measures equal the plain split
    selection's (CMI_RANK_NO_PRUNE=1) ENTRY FOR ENTRY -- trained models, candidate counts across several 64-tile chunks (> 4 096), small
    batches, lists as long as 64, all-tied models (every tile's bound equals the threshold: nothing may be skipped wrongly) and queries
    with long exclusion lists inside skipped tiles."""
    train, test, orc, inst = _setup(model, k, 0, epochs=2, n_users=120, n_items=n_items, n=14000, seed=13)
    kw = dict(bin_thold=2.5, num_recs=num_recs, with_lists=True)
    for batch in ("23", None):
        pruned = _with_env({"CMI_RANK_NO_PRUNE": None, "CMI_RANK_BATCH": batch}, lambda: inst.eval_rankings(_arrays(train), _arrays(test), **kw))
        plain = _with_env({"CMI_RANK_NO_PRUNE": "1", "CMI_RANK_BATCH": batch}, lambda: inst.eval_rankings(_arrays(train), _arrays(test), **kw))
        assert pruned[0]["n_queries"] == plain[0]["n_queries"] > 0 and set(pruned[1]) == set(plain[1])
        for key, lst in plain[1].items():
            assert pruned[1][key] == lst, key          # (item, score) pairs: identical, in order
        for m, v in plain[0].items():
            assert (math.isnan(v) and math.isnan(pruned[0][m])) or pruned[0][m] == v, m
    # an all-tied model (ties resolve in candidate order; with every bound EQUAL to the threshold the `>` must not skip a tile early)
    st = {n: np.zeros_like(a) for n, a in inst.get_states(np.float32).items()}
    inst.set_states(st)
    pruned = _with_env({"CMI_RANK_NO_PRUNE": None}, lambda: inst.eval_rankings(_arrays(train), _arrays(test), **kw))
    plain = _with_env({"CMI_RANK_NO_PRUNE": "1"}, lambda: inst.eval_rankings(_arrays(train), _arrays(test), **kw))
    assert pruned[1] == plain[1] and all((math.isnan(v) and math.isnan(pruned[0][m])) or pruned[0][m] == v for m, v in plain[0].items())


def test_two_instances_evaluate_rankings_at_the_same_time():
    """`cv -p on` evaluates the folds' rankings from one host thread per fold (CARSKit.java:395-412, Recommender.java:1162-1171): two
    instances -- different models, different data, split form and per-query form -- evaluate concurrently, repeatedly, and every result
    equals the instance's lone result entry for entry (per-instance workspace, streams and plan cache; the host's ranged work shares one
    pool)."""
    import threading
    a = _setup("CAMF_CI", 32, 0, epochs=2, n_users=150, n_items=3000, n=15000, seed=31)
    b = _setup("CAMF_CUCI", 10, F64 | STRICT, epochs=1, n_users=80, n_items=400, n=6000, seed=32)
    kw = dict(bin_thold=2.5, num_recs=10, with_lists=True)
    lone = [x[3].eval_rankings(_arrays(x[0]), _arrays(x[1]), **kw) for x in (a, b)]
    errors, done = [], []

    def work(x, want, batch):
        try:
            for rep in range(40):
                os.environ["CMI_RANK_BATCH"] = batch          # (read per call; both threads set small batches: many launches interleave)
                got = x[3].eval_rankings(_arrays(x[0]), _arrays(x[1]), **kw)
                assert got[1] == want[1], rep
                assert all((math.isnan(v) and math.isnan(got[0][m])) or got[0][m] == v for m, v in want[0].items()), rep
            done.append(1)
        except Exception as exc:                              # noqa: BLE001 -- surfaced below
            errors.append(exc)
    stop = threading.Event()

    def trainer():
        # a third fold still TRAINING: fresh instances whose first epoch captures the level graph (thread-local capture) and whose
        # set_ratings / close allocate and free device memory while the other two evaluate
        try:
            d = synth.generate(400, 120, 3, 3, 20000, seed=33)
            st = synth.init_state("CAMF_CI", d, 64, seed=9, dtype=np.float32)
            while not stop.is_set():
                inst = capi.Instance("CAMF_CI", 64, d.n_users, d.n_items, d.n_conds, flags=capi.FLAG_SCHED_CHAIN)
                inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, 3.0)
                inst.set_ratings(d.u, d.j, d.ctx, d.r, d.ctx_ptr, d.ctx_conds)
                inst.set_states(st)
                for _ in range(3):
                    inst.train_epoch(util.LR)
                inst.close()
        except Exception as exc:                              # noqa: BLE001
            errors.append(exc)
    try:
        ts = [threading.Thread(target=work, args=(x, w, "29")) for x, w in zip((a, b), lone)]
        tr = threading.Thread(target=trainer)
        tr.start()
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=300)
        stop.set()
        tr.join(timeout=120)
    finally:
        stop.set()
        os.environ.pop("CMI_RANK_BATCH", None)
    assert not errors, errors
    assert len(done) == 2

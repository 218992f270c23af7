"""Host-only checks of the hub-chain level schedule (carskit_amd/csrc/level_schedule.cpp, build_chain_schedule) -- the order
the default GPU path executes.  The invariants below are exactly what makes "levels in sequence, units of a level in
parallel, a unit's tuples in order" equal to the reference's sequential walk (CAMF_CI.java:80 `for (MatrixEntry me : ...)`):
  * perm is a permutation; units partition it; a unit holds <= max_chain tuples of ONE hub row, consecutive in that row's
    CRS chain, with pairwise distinct spoke rows;
  * two tuples of one level that share a user or an item are in the same unit;
  * for every user and every item, (level, unit, position) order == CRS order.
Plus an end-to-end check on the CPU: replaying the oracle's single-tuple update in schedule order reproduces the sequential
epoch bit for bit."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from carskit_amd import capi, synth

from tests.util import LR, REG, REGC


def _check_chain(u, j, nu, ni, hub, max_chain):
    perm, unit_off, level_off, hub_item = capi.chain_schedule(u, j, nu, ni, hub, max_chain)
    n = len(u)
    assert sorted(perm.tolist()) == list(range(n))
    assert unit_off[0] == 0 and unit_off[-1] == n
    assert level_off[0] == 0 and level_off[-1] == len(unit_off) - 1
    if hub >= 0:
        assert hub_item == bool(hub)
    hubk, spoke = (j, u) if hub_item else (u, j)
    order_key = np.empty((n, 3), dtype=np.int64)  # (level, unit, pos) of every CRS tuple
    for l in range(len(level_off) - 1):
        seen_u, seen_j = {}, {}
        assert level_off[l + 1] > level_off[l]
        lens = []
        for q in range(level_off[l], level_off[l + 1]):
            seg = perm[unit_off[q]:unit_off[q + 1]]
            lens.append(len(seg))
            assert 1 <= len(seg) <= max_chain
            assert len(set(hubk[seg].tolist())) == 1                 # one hub row
            assert len(set(spoke[seg].tolist())) == len(seg)         # distinct spokes
            assert np.all(np.diff(seg) > 0)                          # CRS order inside the unit
            # consecutive in the hub's chain: no other tuple of that hub lies between them
            h = int(hubk[seg[0]])
            chain = np.flatnonzero(hubk == h)
            pos = np.searchsorted(chain, seg)
            assert np.all(np.diff(pos) == 1)
            for p, t in enumerate(seg):
                order_key[t] = (l, q, p)
                for seen, key in ((seen_u, int(u[t])), (seen_j, int(j[t]))):
                    assert seen.setdefault(key, q) == q              # sharing a row inside a level => same unit
        assert lens == sorted(lens, reverse=True)                    # longest units first
    for key in (u, j):
        last = {}
        for t in range(n):
            k = int(key[t])
            if k in last:
                assert tuple(order_key[t]) > tuple(order_key[last[k]])
            last[k] = t
    return perm, unit_off, level_off, hub_item


@pytest.mark.parametrize("hub", [-1, 0, 1])
@pytest.mark.parametrize("max_chain", [1, 2, 16])
def test_chain_schedule_small(hub, max_chain):
    d = synth.generate(37, 13, 2, 3, 600, seed=5)
    perm, unit_off, level_off, _ = _check_chain(d.u, d.j, d.n_users, d.n_items, hub, max_chain)
    if max_chain == 1:  # degenerates to the plain level schedule
        _, off = capi.level_schedule(d.u, d.j, d.n_users, d.n_items, 0)
        assert len(level_off) == len(off)
        assert np.array_equal(np.diff(level_off), np.diff(off))


@settings(max_examples=60, deadline=None)
@given(nu=st.integers(1, 9), ni=st.integers(1, 9), n=st.integers(0, 90), seed=st.integers(0, 1000), hub=st.integers(-1, 1),
       max_chain=st.integers(1, 16))
def test_chain_schedule_property(nu, ni, n, seed, hub, max_chain):
    rng = np.random.default_rng(seed)
    u = rng.integers(0, nu, n).astype(np.int32)
    j = rng.integers(0, ni, n).astype(np.int32)
    _check_chain(u, j, nu, ni, hub, max_chain)


def test_chain_schedule_never_has_more_levels_than_the_plain_one_and_chains_on_c3_like_data():
    d = synth.generate_fast(20000, 2000, 4, 8, 1_000_000, seed=3)
    _, off = capi.level_schedule(d.u, d.j, d.n_users, d.n_items, 0)
    perm, unit_off, level_off, hub_item = capi.chain_schedule(d.u, d.j, d.n_users, d.n_items, -1, 16)
    assert hub_item                                   # items are the busy side (500 ratings each vs 50 per user)
    assert len(level_off) < 0.5 * len(off)            # 941 -> 285 at full C3 size
    assert (len(d.u) / (len(unit_off) - 1)) > 3.0     # mean unit length


def test_schedule_order_replay_equals_sequential_epoch_bitwise():
    """Oracle replay: applying the oracle's update tuple by tuple in CHAIN-SCHEDULE order gives the sequential epoch's model
    bit for bit (fp64), for a model with biases on both sides."""
    from oracle import oracle_c
    d = synth.generate(60, 25, 2, 3, 1500, seed=11)
    k = 6
    gm = oracle_c.global_mean(d.r)
    for model in ("CAMF_CUCI", "CAMF_CI", "CAMF_CU"):
        state = synth.init_state(model, d, k, seed=3)
        mk = lambda u, j, c, r: oracle_c.Oracle(model, k, d.n_users, d.n_items, d.n_conds, u, j, c, r, d.ctx_ptr, d.ctx_conds,
                                                {n: a.copy() for n, a in state.items()}, gm, REG, REG, REG, REGC)
        seq = mk(d.u, d.j, d.ctx, d.r)
        seq.epoch(LR)
        for hub in (0, 1):
            perm, _, _, _ = capi.chain_schedule(d.u, d.j, d.n_users, d.n_items, hub, 16)
            rep = mk(d.u[perm], d.j[perm], d.ctx[perm], d.r[perm])
            rep.epoch(LR)
            for name, a in seq.state.items():
                if a is not None:
                    assert np.array_equal(a, rep.state[name]), (model, hub, name)


def test_arena_positions_thread_every_row_through_its_tuples_in_order():
    """The spoke arena's bookkeeping (cmi_arena_positions): following next[] from first[row] visits exactly the row's stream positions in
    ascending order and returns to first[row]; rows without tuples have first = -1.  Simulating an epoch on it -- every position READS its
    own slot and WRITES slot next[p] -- each row's value must pass through all of its tuples in order and end in first[row]."""
    rng = np.random.default_rng(3)
    for n, n_spokes in ((0, 3), (1, 1), (200, 7), (5000, 900), (3000, 5000)):
        spoke = rng.integers(0, n_spokes, n).astype(np.int32)
        nxt, first = capi.arena_positions(spoke, n_spokes)
        for r in range(n_spokes):
            pos = np.flatnonzero(spoke == r)
            if len(pos) == 0:
                assert first[r] == -1
                continue
            assert first[r] == pos[0]
            walk, p = [], int(first[r])
            for _ in range(len(pos)):
                walk.append(p)
                p = int(nxt[p])
            assert walk == pos.tolist() and p == first[r]
        # one simulated epoch: slot values are (row, number of updates applied so far)
        slot = {int(first[r]): (r, 0) for r in range(n_spokes) if first[r] >= 0}
        for p in range(n):
            row, cnt = slot.pop(p)                     # the tuple at p finds its row's live value in its own slot
            assert row == spoke[p]
            slot[int(nxt[p])] = (row, cnt + 1)
        for r in range(n_spokes):
            if first[r] >= 0:
                assert slot[int(first[r])] == (r, int(np.sum(spoke == r)))
    with pytest.raises(capi.CmiError):
        capi.arena_positions(np.array([0, 5], np.int32), 3)


def test_arena_positions_large_form_matches_a_sorted_reference():
    """From 2^22 tuples on, cmi_arena_positions walks backwards per BUCKET of rows (lists per (range, bucket), answers copied back per
    range of positions) instead of once over all tuples: the same next / first arrays as a stable sort by row gives, for few and many
    rows, with the host threads forced to 5 and left at their default."""
    import ctypes as C
    import os
    from carskit_amd.capi import _p
    L = capi.lib()
    rng = np.random.default_rng(5)
    for n, ns, threads in ((4_300_000, 300_000, "5"), (4_200_000 + 7, 9, None), (4_400_000, 1_500_000, "16")):
        sp = rng.integers(0, ns, n).astype(np.int32)
        nxt, first = np.empty(n, np.int32), np.empty(ns, np.int32)
        if threads:
            os.environ["CMI_HOST_THREADS"] = threads
        try:
            assert L.cmi_arena_positions(n, _p(sp), ns, _p(nxt), _p(first)) == 0
        finally:
            os.environ.pop("CMI_HOST_THREADS", None)
        order = np.argsort(sp, kind="stable")
        so = sp[order]
        starts = np.r_[True, so[1:] != so[:-1]]
        same_next = np.r_[so[1:] == so[:-1], False]
        first_pos = order[starts]
        want_next = np.empty(n, np.int64)
        want_next[order] = np.where(same_next, np.r_[order[1:], 0], first_pos[np.cumsum(starts) - 1])
        want_first = np.full(ns, -1, np.int64)
        want_first[so[starts]] = first_pos
        assert np.array_equal(nxt, want_next.astype(np.int32)) and np.array_equal(first, want_first.astype(np.int32)), (n, ns)

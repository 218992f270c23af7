"""The owner (dataflow) schedule on the host (level_schedule.cpp, build_owner_schedule; no GPU): what sgd_owner relies on.

  * every hub row's tuples sit in ONE owner's list, every list is in CRS order (so the head of some list is always runnable);
  * want[pos] = how many earlier tuples of the epoch share the tuple's spoke row (the tag its record must carry);
  * the flags say exactly when a row can be taken over in registers / must be re-read late / goes back to HBM;
  * executing the lists in ANY interleaving that respects the tags reproduces the sequential epoch bit for bit.
"""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from carskit_amd import capi, synth


def _data(zipf, seed, n_users=300, n_items=80, n=6000):
    d = synth.generate(n_users, n_items, 2, 3, n, seed=seed, item_zipf=zipf)
    return d.u.astype(np.int32), d.j.astype(np.int32), d.n_users, d.n_items


@pytest.mark.parametrize("zipf", [None, 1.1])
@pytest.mark.parametrize("hub", [0, 1])
@pytest.mark.parametrize("n_owners", [1, 7, 64, 1000])
def test_lists_tags_and_flags(zipf, hub, n_owners):
    u, j, nu, ni = _data(zipf, 5)
    depth = 8
    perm, off, want, flags, hub_item = capi.owner_schedule(u, j, nu, ni, n_owners, hub=hub, depth=depth)
    assert hub_item == bool(hub)
    n = len(u)
    assert sorted(perm.tolist()) == list(range(n))
    assert off[0] == 0 and off[-1] == n and np.all(np.diff(off) >= 0)
    hv, sv = (j, u) if hub else (u, j)
    owner_of = {}
    for w in range(n_owners):
        lst = perm[off[w]:off[w + 1]]
        assert np.all(np.diff(lst) > 0)                      # CRS order inside a list
        for t in lst:
            assert owner_of.setdefault(int(hv[t]), w) == w   # a hub row has one owner
    seen = np.zeros(max(nu, ni), dtype=np.int64)
    expect = np.empty(n, dtype=np.int64)
    for t in range(n):
        expect[t] = seen[sv[t]]
        seen[sv[t]] += 1
    assert np.array_equal(want.astype(np.int64), expect[perm])
    F = capi
    for w in range(n_owners):
        b, e = int(off[w]), int(off[w + 1])
        last = {}
        for pos in range(b, e):
            t = perm[pos]
            f = int(flags[pos])
            lp = last.get(int(hv[t]))
            assert bool(f & F.OWN_HUB_FWD) == (lp is not None and pos - lp == 1)
            assert bool(f & F.OWN_HUB_LATE) == (lp is not None and 2 <= pos - lp <= depth)
            last[int(hv[t])] = pos
            fwd = pos > b and sv[perm[pos - 1]] == sv[t] and want[pos] == want[pos - 1] + 1
            assert bool(f & F.OWN_SPK_FWD) == bool(fwd)
            nxt = int(flags[pos + 1]) if pos + 1 < e else 0
            assert bool(f & F.OWN_HUB_STORE) == (not nxt & F.OWN_HUB_FWD)
            assert bool(f & F.OWN_SPK_STORE) == (not nxt & F.OWN_SPK_FWD)


def test_hottest_rows_sit_alone_and_loads_balance():
    u, j, nu, ni = _data(1.1, 9, n_users=2000, n_items=500, n=60000)
    perm, off, want, flags, hub_item = capi.owner_schedule(u, j, nu, ni, 64)
    assert hub_item                                           # items carry the heavy tail
    deg = np.bincount(j, minlength=ni)
    loads = np.diff(off)
    assert loads.max() == deg.max()                           # the hottest item alone defines the busiest owner
    hot_owner = int(np.argmax(loads))
    assert len(set(j[perm[off[hot_owner]:off[hot_owner + 1]]].tolist())) == 1
    rest = np.sort(loads)[:-8]
    assert rest.max() <= 1.5 * max(1, rest.mean()) + deg[np.argsort(deg)[-9]]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_any_tag_respecting_interleaving_is_the_sequential_epoch(seed):
    """A toy non-commutative 'update' (row <- f(row_u, row_j, t)) run (a) sequentially and (b) owner by owner in a random
    interleaving that only ever runs a list head whose spoke tag matches: same final rows, and the run never gets stuck."""
    u, j, nu, ni = _data(1.1, 20 + seed, n_users=120, n_items=40, n=3000)
    n = len(u)

    def upd(a, b, t):
        return (a * 31 + b * 17 + t) % 1000003, (b * 29 + a * 13 + 7 * t) % 1000033

    P, Q = np.arange(nu, dtype=np.int64), np.arange(ni, dtype=np.int64) + 1000
    for t in range(n):
        P[u[t]], Q[j[t]] = upd(int(P[u[t]]), int(Q[j[t]]), t)
    rng = np.random.default_rng(seed)
    for hub in (0, 1):
        perm, off, want, flags, _ = capi.owner_schedule(u, j, nu, ni, 13, hub=hub)
        P2, Q2 = np.arange(nu, dtype=np.int64), np.arange(ni, dtype=np.int64) + 1000
        tag = np.zeros(nu if hub else ni, dtype=np.int64)
        head = off[:-1].copy()
        done = 0
        while done < n:
            ready = [w for w in range(13) if head[w] < off[w + 1] and
                     tag[(u if hub else j)[perm[head[w]]]] == want[head[w]]]
            assert ready, "stuck"
            w = int(rng.choice(ready))
            t = int(perm[head[w]])
            P2[u[t]], Q2[j[t]] = upd(int(P2[u[t]]), int(Q2[j[t]]), t)
            tag[(u if hub else j)[t]] += 1
            head[w] += 1
            done += 1
        assert np.array_equal(P, P2) and np.array_equal(Q, Q2)


def _random_tag_respecting_order(u, j, nu, ni, n_owners, hub, rng):
    """A global execution order that the owner epoch could produce: repeatedly pick, at random, an owner whose head entry's spoke
    row carries the wanted update count, and run that entry.  Returns the CRS indices in execution order."""
    perm, off, want, flags, hub_item = capi.owner_schedule(u, j, nu, ni, n_owners, hub=hub)
    spoke = u if hub_item else j
    tag = np.zeros(max(nu, ni), dtype=np.int64)
    head = off[:-1].copy()
    order = []
    while len(order) < len(u):
        ready = [w for w in range(n_owners) if head[w] < off[w + 1] and tag[spoke[perm[head[w]]]] == want[head[w]]]
        assert ready, "stuck"
        w = int(rng.choice(ready))
        t = int(perm[head[w]])
        order.append(t)
        tag[spoke[t]] += 1
        head[w] += 1
    return np.asarray(order, dtype=np.int64)


@settings(max_examples=60, deadline=None)
@given(nu=st.integers(1, 9), ni=st.integers(1, 9), n=st.integers(0, 90), seed=st.integers(0, 1000), hub=st.integers(-1, 1),
       n_owners=st.integers(1, 12), depth=st.integers(1, 16))
def test_owner_schedule_property(nu, ni, n, seed, hub, n_owners, depth):
    rng = np.random.default_rng(seed)
    u = rng.integers(0, nu, n).astype(np.int32)
    j = rng.integers(0, ni, n).astype(np.int32)
    perm, off, want, flags, hub_item = capi.owner_schedule(u, j, nu, ni, n_owners, hub=hub, depth=depth)
    assert sorted(perm.tolist()) == list(range(n)) and off[0] == 0 and off[-1] == n
    hv, sv = (j, u) if hub_item else (u, j)
    owner_of = {}
    for w in range(n_owners):
        lst = perm[off[w]:off[w + 1]]
        assert np.all(np.diff(lst) > 0)
        for t in lst:
            assert owner_of.setdefault(int(hv[t]), w) == w
    seen = {}
    for t in range(n):                                   # want = rank of the tuple in its spoke row's CRS chain
        pos = int(np.nonzero(perm == t)[0][0])
        assert want[pos] == seen.get(int(sv[t]), 0)
        seen[int(sv[t])] = seen.get(int(sv[t]), 0) + 1
    if n:                                                # and any tag-respecting interleaving runs to the end
        order = _random_tag_respecting_order(u, j, nu, ni, n_owners, hub, rng)
        for key in (u, j):                               # ... keeping every row's tuples in CRS order
            for x in np.unique(key):
                ts = order[key[order] == x]
                assert np.all(np.diff(ts) > 0)


def test_oracle_replay_in_owner_execution_order_equals_sequential_epoch_bitwise():
    """Oracle replay: the oracle's single-tuple update applied in an execution order of the owner epoch (random tag-respecting
    interleaving of the owners' lists) gives the sequential epoch's model bit for bit (fp64), for models with state on both sides."""
    from oracle import oracle_c
    from tests.util import LR, REG, REGC
    d = synth.generate(60, 25, 2, 3, 1500, seed=11, item_zipf=1.2)
    k = 6
    gm = oracle_c.global_mean(d.r)
    rng = np.random.default_rng(5)
    for model in ("CAMF_CUCI", "CAMF_CI", "CAMF_CU", "BiasedMF"):
        state = synth.init_state(model, d, k, seed=3)
        mk = lambda u, j, c, r: oracle_c.Oracle(model, k, d.n_users, d.n_items, d.n_conds, u, j, c, r, d.ctx_ptr, d.ctx_conds,
                                                {n: a.copy() for n, a in state.items()}, gm, REG, REG, REG, REGC)
        seq = mk(d.u, d.j, d.ctx, d.r)
        seq.epoch(LR)
        for hub in (0, 1):
            for n_owners in (3, 40):
                order = _random_tag_respecting_order(d.u, d.j, d.n_users, d.n_items, n_owners, hub, rng)
                rep = mk(d.u[order], d.j[order], d.ctx[order], d.r[order])
                rep.epoch(LR)
                for name, a in seq.state.items():
                    if a is not None:
                        assert np.array_equal(a, rep.state[name]), (model, hub, n_owners, name)

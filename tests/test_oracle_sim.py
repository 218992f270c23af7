"""The oracle for SVD++ / CAMF_ICS / CAMF_LCS / CAMF_MCS (oracle/carskit_oracle_sim.c): bit-for-bit agreement with the second
restatement oracle/oracle_np_sim.py on random small problems, and hand-derived single-update known answers in exact rational
arithmetic (all inputs dyadic and chosen so that every sqrt and division is exact)."""
from fractions import Fraction as F

import numpy as np
import pytest

from oracle import oracle_c, oracle_np_sim as ref

REG, REGC, LR = 1.0 / 8, 1.0 / 16, 1.0 / 256


def _problem(seed, n_users=7, n_items=6, dims=(3, 2), n=60):
    """random tuples; every dimension's LAST condition is its ':na' condition (EmptyContextConditions)"""
    rng = np.random.default_rng(seed)
    n_conds = sum(dims)
    base = np.cumsum([0] + list(dims[:-1]))
    empty = [int(b + d - 1) for b, d in zip(base, dims)]
    combos = [(a, b) for a in range(dims[0]) for b in range(dims[1])]
    conds = [[int(base[0] + a), int(base[1] + b)] for a, b in combos]
    u = rng.integers(0, n_users, n)
    j = rng.integers(0, n_items, n)
    ctx = rng.integers(0, len(conds), n)
    order = np.lexsort((ctx, j, u))
    r = rng.integers(1, 6, n).astype(np.float64)
    ctx_ptr = np.arange(len(conds) + 1, dtype=np.int32) * 2
    ctx_conds = np.array(conds, dtype=np.int32).reshape(-1)
    return dict(n_users=n_users, n_items=n_items, n_conds=n_conds, u=u[order], j=j[order], ctx=ctx[order], r=r[order],
                conds=conds, empty=empty, ctx_ptr=ctx_ptr, ctx_conds=ctx_conds, n_dims=len(dims))


def _lists(a):
    return np.array(a, dtype=np.float64).tolist()


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("model", ["CAMF_ICS", "CAMF_LCS", "CAMF_MCS"])
def test_sim_models_c_equals_python_bitwise(model, seed):
    pr = _problem(seed)
    rng = np.random.default_rng(100 + seed)
    k, numF = 5, 4
    P, Q = rng.random((pr["n_users"], k)), rng.random((pr["n_items"], k))
    st = {"P": P.copy(), "Q": Q.copy()}
    args = (k, pr["u"].tolist(), pr["j"].tolist(), pr["ctx"].tolist(), pr["r"].tolist(), pr["conds"], pr["empty"], _lists(P), _lists(Q),
            0.0, REG, REG, REGC)
    if model == "CAMF_ICS":
        S = np.ones((pr["n_conds"], pr["n_conds"]))
        st["ccMatrix"] = S.copy()
        py = ref.ICS(*args, S=_lists(S))
        name, attr = "ccMatrix", "S"
    elif model == "CAMF_LCS":
        Cm = rng.random((pr["n_conds"], numF))
        st["cfMatrix"] = Cm.copy()
        py = ref.LCS(*args, C=_lists(Cm))
        name, attr = "cfMatrix", "C"
    else:
        x = rng.random(pr["n_conds"]) / np.sqrt(pr["n_dims"])
        if seed == 0:
            x[:] = 0.25          # all positions equal: dist == 0 -> the lowbound branch
        st["cVector"] = x.copy()
        py = ref.MCS(*args, x=_lists(x), n_dims=pr["n_dims"])
        name, attr = "cVector", "x"
    c = oracle_c.SimOracle(model, k, pr["n_users"], pr["n_items"], pr["n_conds"], pr["u"], pr["j"], pr["ctx"], pr["r"], pr["ctx_ptr"],
                           pr["ctx_conds"], pr["empty"], st, 0.0, REG, REG, REG, REGC, n_ctx_dims=pr["n_dims"])
    for _ in range(3):
        lc, lp = c.epoch(LR), py.epoch(LR)
        assert np.isfinite(lc) and lc == lp
    assert np.array_equal(c.state["P"], np.array(py.P)) and np.array_equal(c.state["Q"], np.array(py.Q))
    assert np.array_equal(c.state[name], np.array(getattr(py, attr)))
    assert c.predict(int(pr["u"][0]), int(pr["j"][0]), int(pr["ctx"][0])) == py.predict(int(pr["u"][0]), int(pr["j"][0]), int(pr["ctx"][0]))


@pytest.mark.parametrize("seed", range(4))
def test_svdpp_c_equals_python_bitwise(seed):
    rng = np.random.default_rng(seed)
    nu, ni, k = 6, 9, 4
    pairs = sorted({(int(a), int(b)) for a, b in zip(rng.integers(0, nu, 40), rng.integers(0, ni, 40))})   # 2-D matrix: CRS order
    u = np.array([p[0] for p in pairs], dtype=np.int32)
    j = np.array([p[1] for p in pairs], dtype=np.int32)
    r = rng.integers(1, 6, len(pairs)).astype(np.float64)
    P, Q, Y = rng.standard_normal((nu, k)) * .1, rng.standard_normal((ni, k)) * .1, rng.standard_normal((ni, k)) * .1
    bu, bj = rng.standard_normal(nu) * .1, rng.standard_normal(ni) * .1
    st = {"P": P.copy(), "Q": Q.copy(), "Y": Y.copy(), "userBias": bu.copy(), "itemBias": bj.copy()}
    c = oracle_c.SimOracle("SVD++", k, nu, ni, 0, u, j, None, r, None, None, None, st, 3.25, REG, REG, REG, REGC)
    py = ref.SVDPP(k, nu, u.tolist(), j.tolist(), r.tolist(), _lists(P), _lists(Q), _lists(bu), _lists(bj), _lists(Y), 3.25, REG, REG, REG)
    for _ in range(3):
        lc, lp = c.epoch(LR), py.epoch(LR)
        assert np.isfinite(lc) and lc == lp
    for name, got in (("P", py.P), ("Q", py.Q), ("Y", py.Y), ("userBias", py.bu), ("itemBias", py.bj)):
        assert np.array_equal(c.state[name], np.array(got)), name
    assert c.predict(int(u[0]), int(j[0])) == py.predict(int(u[0]), int(j[0]))


def test_svdpp_single_update_known_answer():
    """One user with 4 rated items (w = sqrt 4 = 2, exact), k = 1, the first tuple of the epoch; everything dyadic."""
    u = np.zeros(4, np.int32)
    j = np.arange(4, dtype=np.int32)
    r = np.array([4.0, 0, 0, 0])
    gm, bu, bj, p, q = F(3), F(1, 2), F(1, 4), F(1, 2), F(1, 2)
    y = [F(1, 2), F(1, 4), F(1, 8), F(1, 8)]
    reg, lr = F(1, 8), F(1, 4)
    w = F(2)
    pred = gm + bu + bj + p * q + sum(yk * q / w for yk in y)
    e = 4 - pred
    want_bu, want_bj = bu + lr * (e - reg * bu), bj + lr * (e - reg * bj)
    s = sum(y) / w
    want_p, want_q = p + lr * (e * q - reg * p), q + lr * (e * (p + s) - reg * q)
    want_y = [yk + lr * (e * q / w - reg * yk) for yk in y]
    st = {"P": np.array([[float(p)]]), "Q": np.array([[float(q)], [0.0], [0.0], [0.0]]), "Y": np.array([[float(v)] for v in y]),
          "userBias": np.array([float(bu)]), "itemBias": np.array([float(bj), 0, 0, 0])}
    c = oracle_c.SimOracle("SVD++", 1, 1, 4, 0, u[:1], j[:1], None, r[:1], None, None, None, st, float(gm), float(reg), float(reg),
                           float(reg), 0.0)
    # the user's item list comes from the tuples given: rebuild the 4-item cache by hand
    c.ui_ptr, c.ui_items = np.array([0, 4], np.int32), np.arange(4, dtype=np.int32)
    c.p.ui_ptr, c.p.ui_items = c.ui_ptr.ctypes.data, c.ui_items.ctypes.data
    loss = c.epoch(float(lr))
    assert c.state["userBias"][0] == float(want_bu) and c.state["itemBias"][0] == float(want_bj)
    assert c.state["P"][0, 0] == float(want_p) and c.state["Q"][0, 0] == float(want_q)
    assert c.state["Y"][:, 0].tolist() == [float(v) for v in want_y]
    want_loss = F(1, 2) * (e * e + reg * bu * bu + reg * bj * bj + reg * p * p + reg * q * q + sum(reg * yk * yk for yk in y))
    assert loss == float(want_loss)


def test_mcs_single_update_known_answer():
    """Two dimensions, differences 3/16 and 4/16 -> dist = 5/16 exactly; k = 1."""
    conds, empty = [[0, 2]], [1, 3]
    x = [F(7, 16), F(4, 16), F(6, 16), F(2, 16)]      # diffs 3/16, 4/16
    p, q, reg, regc, lr, up = F(2), F(1), F(1, 8), F(1, 16), F(1, 8), 1 / np.sqrt(2)
    dist = F(5, 16)
    dot = p * q
    e = F(3) - dot * (1 - dist)
    # e * dot * diff / dist is exact here: (13/8 * 2 * 3/16) / (5/16) is not dyadic -> compare through floats for that term only
    st = {"P": np.array([[float(p)]]), "Q": np.array([[float(q)]]), "cVector": np.array([float(v) for v in x])}
    c = oracle_c.SimOracle("CAMF_MCS", 1, 1, 1, 4, [0], [0], [0], [3.0], [0, 2], [0, 2], empty, st, 0.0, float(reg), float(reg), 0.0,
                           float(regc), n_ctx_dims=2)
    loss = c.epoch(float(lr))
    ef, dotf, distf = float(e), float(dot), float(dist)
    for a, b, d in ((0, 1, 3 / 16), (2, 3, 4 / 16)):
        g = ef * dotf * d / distf
        wa = float(x[a]) + float(lr) * (g - float(regc) * float(x[a]))
        wb = float(x[b]) - float(lr) * (g + float(regc) * float(x[b]))
        wa = 1e-100 if wa < 0 else (up - 1e-100 if wa > up else wa)
        wb = 1e-100 if wb < 0 else (up - 1e-100 if wb > up else wb)
        assert c.state["cVector"][a] == wa and c.state["cVector"][b] == wb
    scale = 1 - dist
    assert c.state["P"][0, 0] == float(p + lr * (e * q * scale - reg * p))
    assert c.state["Q"][0, 0] == float(q + lr * (e * p * scale - reg * q))
    want = F(1, 20) * (e * e + sum(regc * v * v for v in x) + reg * p * p + reg * q * q)      # loss *= 0.05 in the reference
    assert abs(loss - float(want)) <= 1e-15 * float(want)


def test_ics_single_update_known_answer():
    conds, empty = [[0, 2]], [1, 2]                # second dimension: the condition IS its own ':na' -> counts as sim 1, no update
    S = np.ones((3, 3))
    S[0, 1] = S[1, 0] = 0.5
    p, q, reg, regc, lr = F(2), F(1, 2), F(1, 8), F(1, 16), F(1, 4)
    sim = F(1, 2)
    dot = p * q
    e = F(3) - dot * sim
    st = {"P": np.array([[float(p)]]), "Q": np.array([[float(q)]]), "ccMatrix": S}
    c = oracle_c.SimOracle("CAMF_ICS", 1, 1, 1, 3, [0], [0], [0], [3.0], [0, 2], [0, 2], empty, st, 0.0, float(reg), float(reg), 0.0,
                           float(regc))
    loss = c.epoch(float(lr))
    want_s = sim + lr * (e * dot * sim / sim - regc * sim)
    assert c.state["ccMatrix"][0, 1] == float(want_s) == c.state["ccMatrix"][1, 0] and c.state["ccMatrix"][2, 2] == 1.0
    assert c.state["P"][0, 0] == float(p + lr * (e * q * sim - reg * p)) and c.state["Q"][0, 0] == float(q + lr * (e * p * sim - reg * q))
    assert loss == float(F(1, 2) * (e * e + regc * sim * sim + regc * 1 + reg * p * p + reg * q * q))
    assert c.predict(0, 0, 0) == float((p + lr * (e * q * sim - reg * p)) * (q + lr * (e * p * sim - reg * q)) * want_s)

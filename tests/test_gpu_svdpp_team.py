"""SVD++ with a workgroup per chain link (carskit_amd/csrc/svdpp_team.hip): the user's Y rows resident in LDS across the user's ratings.
Held to the sequential C restatement (tree sums: fp64 to rounding, fp32 within the north_star tolerance) on what the kernel has to get
right: every k bucket (factor thread f < k, rows shorter and longer than a 16-lane group), users whose rows exceed the LDS budget (walked
by wave 0 through HBM, in the middle of the epoch), a user split into several runs, a (user, item) pair repeated back to back (the request
for the next rating's Q row would be stale), and agreement with the single-wave kernel it replaces."""
import os

import numpy as np
import pytest

from carskit_amd import capi, synth
from oracle import oracle_c
from tests import util

pytestmark = pytest.mark.gpu
F64, SERIAL = capi.FLAG_STATE_F64, capi.FLAG_SCHED_SERIAL


def _state(nu, ni, k, seed=3):
    rng = np.random.default_rng(seed)
    return {"P": 0.1 * rng.standard_normal((nu, k)), "Q": 0.1 * rng.standard_normal((ni, k)), "userBias": 0.1 * rng.standard_normal(nu),
            "itemBias": 0.1 * rng.standard_normal(ni), "Y": 0.1 * rng.standard_normal((ni, k))}


def _pair(u, j, r, nu, ni, k, flags):
    st = _state(nu, ni, k)
    gm = float(r.mean())
    z = np.zeros(1, np.int32)
    orc = oracle_c.SimOracle("SVD++", k, nu, ni, 1, u, j, None, r, z, np.zeros(0, np.int32), np.zeros(0, np.int32),
                             {n: a.copy() for n, a in st.items()}, gm, util.REG, util.REG, util.REG, util.REGC, n_ctx_dims=1)
    inst = capi.Instance("SVD++", k, nu, ni, 1, flags=flags | SERIAL)
    inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
    inst.set_ratings(u, j, None, r)
    inst.set_states(st)
    return orc, inst


def _matrix(nu, ni, per_user, seed, heavy=()):
    """a 2-D train matrix in row-major order; users in `heavy` rate `heavy[u]` items"""
    rng = np.random.default_rng(seed)
    u, j = [], []
    for x in range(nu):
        m = heavy[x] if x in heavy else int(rng.integers(1, per_user + 1))
        items = np.sort(rng.choice(ni, size=min(m, ni), replace=False))
        u += [x] * len(items)
        j += items.tolist()
    r = rng.integers(1, 6, size=len(u)).astype(np.float64)
    return np.array(u, np.int32), np.array(j, np.int32), r


def _check(orc, inst, flags, epochs=3, lr=util.LR / 4):
    for _ in range(epochs):
        lo, lg = orc.epoch(lr), inst.train_epoch(lr)
        assert abs(lo - lg) <= (1e-11 if flags else 3e-5) * abs(lo)
    for name, a in inst.get_states().items():
        assert np.max(np.abs(orc.state[name].reshape(a.shape) - a)) <= (1e-10 if flags else 3e-4), name


@pytest.mark.parametrize("k", [1, 10, 16, 17, 64, 100, 128, 256, 300])
@pytest.mark.parametrize("flags", [0, F64])
def test_team_matches_the_sequential_restatement(k, flags):
    u, j, r = _matrix(120, 90, 25, seed=k)
    orc, inst = _pair(u, j, r, 120, 90, k, flags)
    _check(orc, inst, flags)


@pytest.mark.parametrize("flags", [0, F64])
def test_users_beyond_the_lds_budget_are_walked_through_hbm(flags):
    """k = 256: 144 KB hold about 138 fp32 rows (69 in fp64); users 3 and 40 rate 400 / 200 items and fall back, their neighbours do not"""
    u, j, r = _matrix(60, 500, 30, seed=5, heavy={3: 400, 40: 200, 41: 150})
    orc, inst = _pair(u, j, r, 60, 500, 256, flags)
    _check(orc, inst, flags, epochs=2)


def test_a_user_in_several_runs_and_a_repeated_pair():
    """Not a train matrix (its (user, item) pairs are unique and row-major), but the C ABI takes any tuple list: the team kernel must then
    do what the single-wave kernel does (the restatement counts a repeated pair twice in |N(u)|, the library's user-items table does not,
    so the two product kernels are compared with each other)."""
    u, j, r = _matrix(40, 60, 20, seed=9)
    # user 7's ratings once more at the end (a second run of the same user), and one pair repeated back to back in the middle
    m = u == 7
    u2 = np.concatenate([u, u[m]])
    j2 = np.concatenate([j, j[m]])
    r2 = np.concatenate([r, r[m][::-1]])
    at = int(np.flatnonzero(u2 == 20)[0])
    u2, j2, r2 = np.insert(u2, at, u2[at]), np.insert(j2, at, j2[at]), np.insert(r2, at, 2.0)
    for flags in (F64, 0):
        _, a = _pair(u2, j2, r2, 40, 60, 64, flags)
        la = [a.train_epoch(util.LR / 4) for _ in range(3)]
        os.environ["CMI_NO_SVDPP_TEAM"] = "1"
        try:
            _, b = _pair(u2, j2, r2, 40, 60, 64, flags)
            lb = [b.train_epoch(util.LR / 4) for _ in range(3)]
        finally:
            del os.environ["CMI_NO_SVDPP_TEAM"]
        np.testing.assert_allclose(la, lb, rtol=1e-11 if flags else 3e-5)
        for name, x in a.get_states().items():
            assert np.max(np.abs(x - b.get_state(name))) <= (1e-10 if flags else 3e-4), name


def test_team_and_the_single_wave_kernel_agree():
    u, j, r = _matrix(200, 150, 30, seed=11)
    _, a = _pair(u, j, r, 200, 150, 64, 0)
    la = [a.train_epoch(util.LR / 4) for _ in range(3)]
    os.environ["CMI_NO_SVDPP_TEAM"] = "1"
    try:
        _, b = _pair(u, j, r, 200, 150, 64, 0)
        lb = [b.train_epoch(util.LR / 4) for _ in range(3)]
    finally:
        del os.environ["CMI_NO_SVDPP_TEAM"]
    np.testing.assert_allclose(la, lb, rtol=2e-5)
    for name, x in a.get_states().items():
        assert np.max(np.abs(x - b.get_state(name))) <= 2e-4, name

"""BASELINE.json's full C3 size (CAMF_CI k=128, 1 M users x 100 K items x 32 conditions, 50 M ratings) on the GPU:
a direct one-epoch comparison with the CPU oracle (about 30 s of single-thread CPU), plus size-independent
properties -- idempotence at lr = 0, loss consistency with evalRatings, and schedule independence (the level
schedule, eager launches and the two-lane graph give the bit-identical model)."""
import numpy as np
import pytest

from carskit_amd import capi, synth
from tests import util

pytestmark = pytest.mark.gpu

K = 128


@pytest.fixture(scope="module")
def c3():
    data = synth.generate_fast(1_000_000, 100_000, 4, 8, 50_000_000)
    state = synth.init_state("CAMF_CI", data, K, dtype=np.float32)
    gm = float(data.r.sum() / np.count_nonzero(data.r))
    return data, state, gm


def _inst(c3, flags=0):
    data, state, gm = c3
    inst = capi.Instance("CAMF_CI", K, data.n_users, data.n_items, data.n_conds, flags=flags)
    inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
    inst.set_ratings(data.u, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
    inst.set_states(state)
    return inst


def test_c3_one_epoch_matches_oracle(c3):
    data, state, gm = c3
    inst = _inst(c3)
    assert inst.schedule_info()["tuples"] == data.n and inst.schedule_info()["levels"] > 500
    lg = inst.train_epoch(util.LR)
    orc = util.c_oracle("CAMF_CI", data, K, {n: a.astype(np.float64) for n, a in state.items()}, gm)
    lo = orc.epoch(util.LR)
    assert abs(lg - lo) <= 1e-6 * abs(lo)                       # 50 M fp32 updates vs fp64, same order
    for name in ("P", "Q", "userBias", "icBias"):
        d = np.abs(inst.get_state(name, np.float64) - orc.state[name].reshape(inst.state_shape(name)))
        assert d.max() <= 2e-5, (name, d.max())
    # RMSE over a 2 M-tuple sample of the training set, clamped like evalRatings: the fp32 bar of the north star
    idx = np.arange(0, data.n, 25)
    ge = inst.eval_ratings(data.u[idx], data.j[idx], data.ctx[idx], data.r[idx], 1.0, 5.0)
    oe = orc.eval_ratings(data.u[idx], data.j[idx], data.ctx[idx], data.r[idx], 1.0, 5.0)
    assert abs(ge["RMSE"] - oe["RMSE"]) <= 1e-5 and abs(ge["MAE"] - oe["MAE"]) <= 1e-5


def test_c3_lr_zero_is_idempotent_and_loss_is_consistent(c3):
    data, state, gm = c3
    inst = _inst(c3)
    loss = inst.train_epoch(0.0)
    for name, a in state.items():
        assert np.array_equal(inst.get_state(name, np.float32), a), name     # nothing moved
    # with lr = 0 the epoch loss is 0.5 * (sum e^2 + regularisers of the visited entries), all at the initial model
    pred = inst.predict(data.u, data.j, data.ctx)
    e2 = float(np.sum((data.r - pred) ** 2))
    P, Q = state["P"].astype(np.float64), state["Q"].astype(np.float64)
    np2, nq2 = (P * P).sum(axis=1), (Q * Q).sum(axis=1)
    bu = state["userBias"].astype(np.float64)
    ic = state["icBias"].astype(np.float64)
    conds = data.ctx_conds.reshape(-1, data.n_dims)[data.ctx]             # [n, D] condition ids (fixed D here)
    reg = (util.REG * np2[data.u].sum() + util.REG * nq2[data.j].sum() + util.REG * (bu[data.u] ** 2).sum()
           + util.REGC * (ic[data.j[:, None], conds] ** 2).sum())
    assert abs(loss - 0.5 * (e2 + reg)) <= 2e-6 * loss


def test_c3_schedules_agree_bit_for_bit(c3):
    outs = []
    for flags in (0, capi.FLAG_NO_GRAPH, capi.FLAG_TWO_LANE):
        inst = _inst(c3, flags)
        losses = [inst.train_epoch(util.LR) for _ in range(2)]
        outs.append((losses, inst.get_state("P", np.float32), inst.get_state("Q", np.float32),
                     inst.get_state("icBias", np.float32)))
        del inst
    for other in outs[1:]:
        np.testing.assert_allclose(other[0], outs[0][0], rtol=1e-12)
        for a, b in zip(outs[0][1:], other[1:]):
            assert np.array_equal(a, b)

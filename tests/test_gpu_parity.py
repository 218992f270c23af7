"""Parity of the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Bars (written where they are asserted):
  * fp64 state + strict order, serial schedule : model state AND per-epoch loss bit-identical
  * fp64 state + strict order, level schedule  : model state bit-identical, loss to 1e-12 relative
  * fp64 state, default (tree-reduced dot)      : RMSE/MAE within 1e-9
  * fp32 state (the throughput configuration)   : RMSE/MAE within 1e-5 (BASELINE.json north_star)
"""
import numpy as np
import pytest

from carskit_amd import capi, synth
from oracle import oracle_c
from tests import util

pytestmark = pytest.mark.gpu

F64, SERIAL, STRICT, NOGRAPH = (capi.FLAG_STATE_F64, capi.FLAG_SCHED_SERIAL, capi.FLAG_STRICT, capi.FLAG_NO_GRAPH)


def make_pair(model, data, k, flags, seed=5, regs=None):
    """(oracle, gpu instance) over the same tuples and the same injected initial state."""
    regU, regI, regB, regC = regs or (util.REG, util.REG, util.REG, util.REGC)
    state = synth.init_state(model, data, k, seed=seed)
    gm = oracle_c.global_mean(data.r)
    orc = util.c_oracle(model, data, k, state, gm, regU, regI, regB, regC)
    u, j, ctx, r = util.tuples_for(model, data)
    inst = capi.Instance(model, k, data.n_users, data.n_items, data.n_conds, flags=flags)
    inst.set_hparams(regU, regI, regB, regC, gm)
    if model in util.TWO_D:
        inst.set_ratings(u, j, None, r)
    else:
        inst.set_ratings(u, j, ctx, r, data.ctx_ptr, data.ctx_conds)
    inst.set_states(state)
    return orc, inst


def assert_state_equal(orc, inst, exact=True, atol=0.0):
    for name, a in inst.get_states().items():
        ref = orc.state[name].reshape(a.shape)
        if exact:
            assert np.array_equal(ref, a), name
        else:
            assert np.max(np.abs(ref - a)) <= atol, (name, np.max(np.abs(ref - a)))


def test_instances_are_reentrant_across_threads():
    """The reference runs one recommender per CV fold on its own Java thread (CARSKit.java:395-412).  Handles carry
    their own stream/graph and the library has no global state: concurrent training from several host threads gives
    exactly the results of running the same instances one after another."""
    import threading
    data = util.small_data(n_users=4000, n_items=500, n_dims=3, conds_per_dim=3, n=90000, seed=71)
    models = ["CAMF_CI", "CAMF_CU", "CAMF_CUCI", "BiasedMF", "PMF"]

    def run(model, out, idx):
        _, inst = make_pair(model, data, 64, 0, seed=idx)
        losses, _ = inst.train(6, util.LR, bold_driver=True)
        out[idx] = (losses.tolist(), inst.get_states(np.float32))

    seq, par = {}, {}
    for i, m in enumerate(models):
        run(m, seq, i)
    ths = [threading.Thread(target=run, args=(m, par, i)) for i, m in enumerate(models)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for i in range(len(models)):
        assert par[i][0] == seq[i][0], models[i]
        for name in seq[i][1]:
            assert np.array_equal(par[i][1][name], seq[i][1][name]), (models[i], name)


@pytest.mark.parametrize("model,k", [("CAMF_CI", 128), ("CAMF_CU", 64), ("BiasedMF", 10)])
def test_f32_holds_1e5_over_the_default_100_epochs(model, k):
    """setting.conf defaults: num.max.iter=100 with the bold driver.  fp32 rounding must not accumulate past the
    north_star's 1e-5, nor flip a bold-driver decision, over the full schedule."""
    data = util.small_data(n_users=2500, n_items=350, n_dims=3, conds_per_dim=4, n=50000, seed=81)
    train, test = synth.split(data, 0.2)
    orc, inst = make_pair(model, train, k, 0)
    o_losses, o_lrs, _ = orc.build_model(100, util.LR, bold_driver=True)
    g_losses, g_lrs = inst.train(100, util.LR, bold_driver=True)
    assert len(g_losses) == len(o_losses)
    # Late in the schedule the bold driver has shrunk lRate by orders of magnitude and consecutive losses differ by
    # less than fp32 resolution, so `abs(last_loss) > abs(loss)` becomes a coin flip between fp32 and fp64; by then a
    # step moves no parameter by a representable amount.  The decisions must agree while the rate still matters.
    live = o_lrs >= 1e-5
    assert live.sum() >= 20 and g_lrs[live].tolist() == o_lrs[live].tolist()
    np.testing.assert_allclose(g_losses, o_losses, rtol=5e-5)
    tctx = None if model in util.TWO_D else test.ctx
    oe = orc.eval_ratings(test.u, test.j, tctx, test.r, 1.0, 5.0)
    ge = inst.eval_ratings(test.u, test.j, tctx, test.r, 1.0, 5.0)
    assert abs(oe["RMSE"] - ge["RMSE"]) <= 1e-5 and abs(oe["MAE"] - ge["MAE"]) <= 1e-5
    # and the fp64 state reproduces the whole 100-epoch trajectory
    _, i64 = make_pair(model, train, k, F64)
    d_losses, d_lrs = i64.train(100, util.LR, bold_driver=True)
    assert d_lrs.tolist() == o_lrs.tolist()
    de = i64.eval_ratings(test.u, test.j, tctx, test.r, 1.0, 5.0)
    assert abs(oe["RMSE"] - de["RMSE"]) <= 1e-9


def test_instance_can_be_reloaded_with_other_ratings():
    """cmi_set_ratings may be called again on the same handle (next fold): the schedule, the device tuple stream and
    the captured graph are rebuilt."""
    d1 = util.small_data(n_users=900, n_items=120, n=15000, seed=91)
    d2 = synth.RatingData(d1.n_users, d1.n_items, d1.n_conds, d1.n_dims, d1.u[::-1].copy(), d1.j[::-1].copy(),
                          d1.ctx[::-1].copy(), d1.r[::-1].copy(), d1.ctx_ptr, d1.ctx_conds)
    _, reused = make_pair("CAMF_CUCI", d1, 64, 0)
    for _ in range(2):
        reused.train_epoch(util.LR)
    st = synth.init_state("CAMF_CUCI", d2, 64, seed=6)
    reused.set_ratings(d2.u, d2.j, d2.ctx, d2.r, d2.ctx_ptr, d2.ctx_conds)
    reused.set_states(st)
    _, fresh = make_pair("CAMF_CUCI", d2, 64, 0, seed=6)
    for _ in range(3):
        assert reused.train_epoch(util.LR) == fresh.train_epoch(util.LR)
    for name, a in fresh.get_states(np.float32).items():
        assert np.array_equal(a, reused.get_state(name, np.float32)), name

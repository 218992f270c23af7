"""SVD++ and CAMF_ICS / CAMF_LCS / CAMF_MCS on the GPU (carskit_amd/csrc/ext_kernels.hip) against the oracle
(oracle/carskit_oracle_sim.c).  Bars:
  * fp64 state + STRICT : model state AND per-epoch loss bit-identical (one lane replays the reference's operation sequence);
  * fp64 state          : state within 1e-11, loss within 1e-12 relative (tree-reduced dots);
  * fp32 state          : predictions on held-out tuples within 2e-4 (these are ranking models: the score only has to order items).
All four need CMI_FLAG_SCHED_SERIAL (every rating updates parameters every other rating reads)."""
import numpy as np
import pytest

from carskit_amd import capi, synth
from oracle import oracle_c
from tests import util

pytestmark = pytest.mark.gpu
F64, SERIAL, STRICT = capi.FLAG_STATE_F64, capi.FLAG_SCHED_SERIAL, capi.FLAG_STRICT
MODELS = ["SVD++", "CAMF_ICS", "CAMF_LCS", "CAMF_MCS"]
NUM_F = 7


def _data(seed=81, n=1500):
    """every dimension's last condition plays its ':na' condition (EmptyContextConditions, DataDAO.java:213-214)"""
    d = util.small_data(n_users=70, n_items=30, n_dims=3, conds_per_dim=4, n=n, seed=seed)
    empty = np.array([dim * 4 + 3 for dim in range(3)], dtype=np.int32)
    return d, empty


def _state(model, d, k, seed=7):
    rng = np.random.default_rng(seed)
    st = {"P": rng.random((d.n_users, k)), "Q": rng.random((d.n_items, k))}      # isRankingPred: P.init(), Q.init() (CAMF_ICS.java:40-46)
    if model == "SVD++":
        st = {"P": 0.1 * rng.standard_normal((d.n_users, k)), "Q": 0.1 * rng.standard_normal((d.n_items, k)),
              "userBias": 0.1 * rng.standard_normal(d.n_users), "itemBias": 0.1 * rng.standard_normal(d.n_items),
              "Y": 0.1 * rng.standard_normal((d.n_items, k))}
    elif model == "CAMF_ICS":
        st["P"] *= 0.3
        st["ccMatrix"] = np.ones((d.n_conds, d.n_conds))
    elif model == "CAMF_LCS":
        st["P"] *= 0.3
        st["cfMatrix"] = rng.random((d.n_conds, NUM_F))
    else:
        st["P"] *= 0.02      # keeps e * dot small, so the positions stay inside (0, upbound) instead of being parked on a bound
        st["cVector"] = (0.2 + 0.6 * rng.random(d.n_conds)) / np.sqrt(d.n_dims)
    return st


def make(model, d, empty, k, flags, lr_state_seed=7):
    st = _state(model, d, k, lr_state_seed)
    gm = oracle_c.global_mean(d.r)
    if model == "SVD++":
        u, j, r = synth.to_2d(d)
        ctx = None
    else:
        u, j, ctx, r = d.u, d.j, d.ctx, d.r
    orc = oracle_c.SimOracle(model, k, d.n_users, d.n_items, d.n_conds, u, j, ctx, r, d.ctx_ptr, d.ctx_conds, empty,
                             {n_: a.copy() for n_, a in st.items()}, gm, util.REG, util.REG, util.REG, util.REGC, n_ctx_dims=d.n_dims)
    inst = capi.Instance(model, k, d.n_users, d.n_items, d.n_conds, flags=flags | SERIAL)
    inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
    inst.set_sim_params(NUM_F, d.n_dims, empty)
    if model == "SVD++":
        inst.set_ratings(u, j, None, r)
    else:
        inst.set_ratings(u, j, ctx, r, d.ctx_ptr, d.ctx_conds)
    inst.set_states(st)
    return orc, inst


LR = util.LR / 8     # multiplicative similarities with U(0,1) factors: the reference's default rate diverges on synthetic data


@pytest.mark.parametrize("model", MODELS)
@pytest.mark.parametrize("k", [5, 10, 70])
def test_strict_f64_bit_exact(model, k):
    d, empty = _data()
    orc, inst = make(model, d, empty, k, F64 | STRICT)
    for _ in range(4):
        lo, lg = orc.epoch(LR), inst.train_epoch(LR)
        assert np.isfinite(lo) and lo == lg
    for name, a in inst.get_states().items():
        assert np.array_equal(orc.state[name].reshape(a.shape), a), name


@pytest.mark.parametrize("model", MODELS)
def test_wave_f64_and_f32(model):
    d, empty = _data(seed=82, n=2500)
    train, test = synth.split(d, 0.2)
    for flags, tol_state, tol_pred in ((F64, 1e-11, 1e-10), (0, 2e-4, 2e-4)):
        orc, inst = make(model, train, empty, 64, flags)
        # (CAMF_MCS's update has a sign discontinuity -- diff / dist with positions that the clipping rule parks on the same
        #  bound -- so once positions saturate, a different ROUNDING of the dot product is a different trajectory and only the
        #  strict kernel can follow the oracle; _state() keeps this problem in the smooth regime)
        for _ in range(5):
            lo, lg = orc.epoch(LR), inst.train_epoch(LR)
            assert abs(lo - lg) <= (1e-12 if flags else 2e-5) * abs(lo)
        for name, a in inst.get_states().items():
            assert np.max(np.abs(orc.state[name].reshape(a.shape) - a)) <= tol_state, name
        tctx = None if model == "SVD++" else test.ctx
        got = inst.predict(test.u, test.j, tctx)
        want = np.array([orc.predict(int(u), int(j), int(c)) for u, j, c in zip(test.u, test.j, test.ctx)])
        assert np.max(np.abs(got - want)) <= tol_pred


def test_serial_flag_and_sim_params_are_required():
    d, empty = _data()
    with pytest.raises(capi.CmiError) as ei:
        capi.Instance("CAMF_ICS", 8, d.n_users, d.n_items, d.n_conds)
    assert ei.value.code == capi.E_UNSUPPORTED
    inst = capi.Instance("CAMF_LCS", 8, d.n_users, d.n_items, d.n_conds, flags=SERIAL)
    with pytest.raises(capi.CmiError):
        inst.set_ratings(d.u, d.j, d.ctx, d.r, d.ctx_ptr, d.ctx_conds)      # EmptyContextConditions / numF not given yet


def test_save_load_roundtrip_of_the_extra_containers(tmp_path):
    d, empty = _data()
    for model in MODELS:
        _, a = make(model, d, empty, 16, F64)
        a.train_epoch(LR)
        a.save_model(tmp_path / "m.cmi")
        _, b = make(model, d, empty, 16, F64, lr_state_seed=99)
        b.load_model(tmp_path / "m.cmi")
        for name, arr in a.get_states().items():
            assert np.array_equal(arr, b.get_states()[name]), (model, name)


@pytest.mark.parametrize("model", MODELS)
def test_rankings_match_oracle_in_fp64(model):
    """cmi_eval_rankings for these models (score = <a_q, b_j> + const_q with a_q = s(c) * P[u], or for SVD++
    a_q = [P[u] + sum Y / sqrt|N(u)| | 1]; ext_kernels.hip) against oracle/rank_oracle.py driven by the oracle's scalar predict():
    identical top-N lists, scores within 1e-10, all measures within 1e-12."""
    from oracle import rank_oracle
    d, empty = _data(seed=83, n=2500)
    train, test = synth.split(d, 0.25)
    orc, inst = make(model, train, empty, 16, F64 | STRICT)
    for _ in range(2):
        orc.epoch(LR)
        inst.train_epoch(LR)
    tup = lambda t: list(zip(t.u.tolist(), t.j.tolist(), t.ctx.tolist(), t.r.tolist()))
    ref, ref_lists = rank_oracle.eval_rankings(lambda u, j, c: orc.predict(u, j, -1 if model == "SVD++" else c), tup(train), tup(test),
                                               bin_thold=-1.0, num_recs=10, strategy="ucu", num_ignore=-1)
    res, lists = inst.eval_rankings((train.u, train.j, train.ctx, train.r), (test.u, test.j, test.ctx, test.r), bin_thold=-1.0,
                                    num_recs=10, num_ignore=-1, strategy="ucu", with_lists=True)
    assert res.pop("n_queries") == len(ref_lists) > 20
    for key, want in ref_lists.items():
        got = lists[key]
        assert [i for i, _ in got] == [i for i, _ in want], key
        assert max(abs(a - b) for (_, a), (_, b) in zip(got, want)) <= 1e-10
    for m, v in ref.items():
        assert abs(res[m] - v) <= 1e-12, m


def test_measure_hbm_reports_plausible_rates():
    copy, rows = capi.measure_hbm(0, 2 << 30)
    assert 1500 < copy < 8000 and 1500 < rows < 8000          # GB/s: a working MI355X sits at 4-6.5 TB/s on both

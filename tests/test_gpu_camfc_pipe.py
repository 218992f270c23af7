"""CAMF_C as one software-pipelined wave (carskit_amd/csrc/camfc_pipe.hip): rows requested D tuples ahead, stale requests replaced from
an LDS ring of the last D tuples' rows.  Order-exact like every CAMF_C path; the dot is a tree sum, so fp64 state agrees with the
sequential oracle to rounding and fp32 state within the north_star tolerance.  The cases force what the pipeline has to get right:
hazards (few items / few users: a requested row rewritten by one of the D tuples in between), user-sorted input (forwarding from
registers), n not a multiple of the 64-tuple chunk, n < 64, every k bucket (masked and full rows), missing conditions."""
import dataclasses
import os

import numpy as np
import pytest

from carskit_amd import capi, synth
from tests import util
from tests.test_gpu_parity import assert_state_equal, make_pair

pytestmark = pytest.mark.gpu
SERIAL, F64 = capi.FLAG_SCHED_SERIAL, capi.FLAG_STATE_F64


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    try:
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _run(data, k, flags, epochs=4):
    # CMI_NO_CAMFC_BLOCKS: the conflict-free-block path would take over on unsorted data; here the serial wave is under test
    def go():
        orc, inst = make_pair("CAMF_C", data, k, SERIAL | flags)
        assert inst.schedule_info()["flow_blocks"] == 0
        o_losses, o_lrs, _ = orc.build_model(epochs, util.LR, bold_driver=True)
        g_losses, g_lrs = inst.train(epochs, util.LR, bold_driver=True)
        return orc, inst, o_losses, o_lrs, g_losses, g_lrs
    return _with_env({"CMI_NO_CAMFC_BLOCKS": "1", "CMI_NO_CAMFC_PIPE": None}, go)


@pytest.mark.parametrize("k", [1, 10, 63, 64, 100, 128, 200, 256])
@pytest.mark.parametrize("flags", [0, F64])
def test_pipe_matches_the_sequential_oracle(k, flags):
    data = util.small_data(n_users=300, n_items=120, n_dims=3, conds_per_dim=4, n=5000 + k, seed=40 + k)
    orc, inst, o_losses, o_lrs, g_losses, g_lrs = _run(data, k, flags)
    assert g_lrs.tolist() == o_lrs.tolist()
    np.testing.assert_allclose(g_losses, o_losses, rtol=1e-11 if flags else 3e-5)
    assert_state_equal(orc, inst, exact=False, atol=1e-10 if flags else 3e-4)


@pytest.mark.parametrize("n_users,n_items,n", [(5, 4, 3000), (3, 200, 2500), (200, 3, 2500), (64, 64, 4096), (2, 2, 700)])
def test_hazards_a_requested_row_rewritten_before_it_is_used(n_users, n_items, n):
    """few users / items: almost every tuple's row was written by one of the D tuples since its request went out"""
    data = util.small_data(n_users=n_users, n_items=n_items, n_dims=2, conds_per_dim=3, n=n, seed=7 + n_users)
    for flags in (F64, 0):
        orc, inst, o_losses, o_lrs, g_losses, g_lrs = _run(data, 64, flags, epochs=3)
        np.testing.assert_allclose(g_losses, o_losses, rtol=1e-11 if flags else 3e-5)
        assert_state_equal(orc, inst, exact=False, atol=1e-10 if flags else 3e-4)


@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 127, 128, 200])
def test_chunk_boundaries(n):
    data = util.small_data(n_users=30, n_items=20, n_dims=2, conds_per_dim=3, n=n, seed=90 + n)
    orc, inst, o_losses, _, g_losses, _ = _run(data, 16, F64, epochs=3)
    np.testing.assert_allclose(g_losses, o_losses, rtol=1e-11)
    assert_state_equal(orc, inst, exact=False, atol=1e-11)


def test_user_sorted_input_and_missing_conditions():
    data = util.small_data(n_users=40, n_items=300, n_dims=4, conds_per_dim=3, n=3000, seed=34)
    order = np.lexsort((data.j, data.u))
    srt = dataclasses.replace(data, u=data.u[order], j=data.j[order], ctx=data.ctx[order], r=data.r[order])
    orc, inst, o_losses, _, g_losses, _ = _run(srt, 128, F64, epochs=3)
    np.testing.assert_allclose(g_losses, o_losses, rtol=1e-11)
    assert_state_equal(orc, inst, exact=False, atol=1e-10)


def test_pipe_and_the_one_ahead_wave_agree():
    """same data through sgd_serial_fast (CMI_NO_CAMFC_PIPE): both are tree-sum kernels over the same order"""
    data = util.small_data(n_users=150, n_items=60, n_dims=3, conds_per_dim=3, n=4000, seed=3)
    _, a, _, _, la, _ = _run(data, 64, 0, epochs=3)

    def old():
        orc, inst = make_pair("CAMF_C", data, 64, SERIAL)
        return inst, inst.train(3, util.LR, bold_driver=True)[0]
    b, lb = _with_env({"CMI_NO_CAMFC_BLOCKS": "1", "CMI_NO_CAMFC_PIPE": "1"}, old)
    np.testing.assert_allclose(la, lb, rtol=2e-5)
    for name, x in a.get_states().items():
        assert np.max(np.abs(x - b.get_state(name))) <= 2e-4, name


def test_camf_c_without_context_dimensions_takes_the_general_paths():
    """dmax = 0 (every rating in the one context that has no condition): the dimension-specialised chains do not apply; = the oracle"""
    from oracle import oracle_c
    rng = np.random.default_rng(5)
    nu, ni, n, k = 50, 40, 900, 16
    u, j = rng.integers(nu, size=n).astype(np.int32), rng.integers(ni, size=n).astype(np.int32)
    ctx, r = np.zeros(n, np.int32), rng.integers(1, 6, size=n).astype(np.float64)
    ctx_ptr, ctx_conds = np.zeros(2, np.int32), np.zeros(0, np.int32)
    for flags in (F64, 0):
        st = {"P": 0.1 * rng.standard_normal((nu, k)), "Q": 0.1 * rng.standard_normal((ni, k)), "userBias": 0.1 * rng.standard_normal(nu),
              "itemBias": 0.1 * rng.standard_normal(ni), "condBias": 0.1 * rng.standard_normal(1)}
        gm = float(r.mean())
        orc = oracle_c.Oracle("CAMF_C", k, nu, ni, 1, u, j, ctx, r, ctx_ptr, ctx_conds, {a_: b_.copy() for a_, b_ in st.items()}, gm, util.REG,
                              util.REG, util.REG, util.REGC)
        inst = capi.Instance("CAMF_C", k, nu, ni, 1, flags=SERIAL | flags)
        inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
        inst.set_ratings(u, j, ctx, r, ctx_ptr, ctx_conds)
        inst.set_states(st)
        for _ in range(2):
            lo, lg = orc.epoch(util.LR), inst.train_epoch(util.LR)
            assert abs(lo - lg) <= (1e-11 if flags else 3e-5) * abs(lo)
        assert_state_equal(orc, inst, exact=False, atol=1e-10 if flags else 3e-4)


@pytest.mark.parametrize("k,n_dims,cpd,flags", [(64, 8, 43, 0), (64, 8, 43, F64), (10, 3, 30, F64), (128, 5, 100, 0), (256, 8, 120, F64), (100, 2, 40, 0)])
def test_pipe_with_more_than_64_conditions_keeps_condbias_in_lds(k, n_dims, cpd, flags):
    """Round 4: the real Frappe file has 343 conditions in 8 dimensions -- beyond the one-condition-per-lane register form.  There condBias
    sits in LDS, the ids travel as 16-bit fields of two packed words, lane d reads / rewrites the d-th condition's entry.  Same order-exact
    chain: = the sequential oracle (fp64: to rounding; fp32: the north_star tolerance)."""
    data = util.small_data(n_users=200, n_items=150, n_dims=n_dims, conds_per_dim=cpd, n=4000 + k, seed=60 + k)
    assert 64 < data.n_conds <= 1024
    orc, inst, o_losses, o_lrs, g_losses, g_lrs = _run(data, k, flags)
    assert g_lrs.tolist() == o_lrs.tolist()
    np.testing.assert_allclose(g_losses, o_losses, rtol=1e-11 if flags else 3e-5)
    assert_state_equal(orc, inst, exact=False, atol=1e-10 if flags else 3e-4)


def test_lds_condbias_form_with_hazards_missing_conditions_and_odd_sizes():
    """few users (every requested row rewritten inside the look-ahead), ragged contexts (some dimensions absent), n not a multiple of 64"""
    base = util.small_data(n_users=4, n_items=90, n_dims=6, conds_per_dim=20, n=2777, seed=77)
    # drop the last condition of every third context: ragged condition lists
    ptr, conds = [0], []
    for c in range(base.n_ctx):
        row = base.ctx_conds[base.ctx_ptr[c]:base.ctx_ptr[c + 1]].tolist()
        if c % 3 == 0:
            row = row[:-1]
        conds += row
        ptr.append(len(conds))
    data = dataclasses.replace(base, ctx_ptr=np.asarray(ptr, np.int32), ctx_conds=np.asarray(conds, np.int32))
    for flags in (F64, 0):
        orc, inst, o_losses, _, g_losses, _ = _run(data, 64, flags, epochs=3)
        np.testing.assert_allclose(g_losses, o_losses, rtol=1e-11 if flags else 3e-5)
        assert_state_equal(orc, inst, exact=False, atol=1e-10 if flags else 3e-4)

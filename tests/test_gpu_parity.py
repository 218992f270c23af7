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


def make_pair(model, data, k, flags, seed=5, regs=None, before_ratings=None):
    """(oracle, gpu instance) over the same tuples and the same injected initial state."""
    regU, regI, regB, regC = regs or (util.REG, util.REG, util.REG, util.REGC)
    state = synth.init_state(model, data, k, seed=seed)
    gm = oracle_c.global_mean(data.r)
    orc = util.c_oracle(model, data, k, state, gm, regU, regI, regB, regC)
    u, j, ctx, r = util.tuples_for(model, data)
    inst = capi.Instance(model, k, data.n_users, data.n_items, data.n_conds, flags=flags)
    inst.set_hparams(regU, regI, regB, regC, gm)
    if before_ratings:
        before_ratings(inst)
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


@pytest.mark.parametrize("model", util.MODELS)
@pytest.mark.parametrize("k", [3, 10, 64, 70])
def test_serial_strict_f64_bit_exact(model, k):
    data = util.small_data(n_users=60, n_items=25, n=900, seed=21)
    orc, inst = make_pair(model, data, k, F64 | SERIAL | STRICT)
    o_losses, o_lrs, _ = orc.build_model(6, util.LR, bold_driver=True)
    g_losses, g_lrs = inst.train(6, util.LR, bold_driver=True)
    assert g_losses.tolist() == o_losses.tolist()      # bit-identical epoch losses
    assert g_lrs.tolist() == o_lrs.tolist()            # hence identical bold-driver trajectory
    assert_state_equal(orc, inst, exact=True)


@pytest.mark.parametrize("model", [m for m in util.MODELS if m != "CAMF_C"])
@pytest.mark.parametrize("k", [5, 64, 130])
@pytest.mark.parametrize("graph", [True, False])
def test_level_strict_f64_state_bit_exact(model, k, graph):
    data = util.small_data(n_users=300, n_items=40, n=4000, seed=22)
    orc, inst = make_pair(model, data, k, F64 | STRICT | (0 if graph else NOGRAPH))
    lr = util.LR
    for _ in range(4):
        lo = orc.epoch(lr)
        lg = inst.train_epoch(lr)
        assert abs(lo - lg) <= 1e-12 * abs(lo)          # loss: same terms, different (fixed) summation tree
    assert_state_equal(orc, inst, exact=True)           # state: the level schedule commutes exactly
    u, j, _, _ = util.tuples_for(model, data)
    assert len(capi.level_schedule(u, j, data.n_users, data.n_items)[1]) - 1 >= np.bincount(j).max()
    assert inst.schedule_info()["levels"] >= 1


@pytest.mark.parametrize("model", [m for m in util.MODELS if m != "CAMF_C"])
def test_level_f64_default(model):
    data = util.small_data(n_users=400, n_items=60, n_dims=3, conds_per_dim=4, n=8000, seed=23)
    train, test = synth.split(data, 0.2)
    orc, inst = make_pair(model, train, 64, F64)
    o_losses, _, _ = orc.build_model(10, util.LR, bold_driver=True)
    g_losses, _ = inst.train(10, util.LR, bold_driver=True)
    np.testing.assert_allclose(g_losses, o_losses, rtol=1e-11)
    tu, tj, tctx, tr = util.tuples_for(model, test) if model not in util.TWO_D else (test.u, test.j, None, test.r)
    oe = orc.eval_ratings(tu, tj, tctx, tr, 1.0, 5.0)
    ge = inst.eval_ratings(tu, tj, tctx, tr, 1.0, 5.0)
    assert oe["n"] == ge["n"]
    for key in ("MAE", "RMSE", "NMAE", "rMAE", "rRMSE"):
        assert abs(oe[key] - ge[key]) <= 1e-9, key       # fp64 bar
    assert_state_equal(orc, inst, exact=False, atol=1e-12)


@pytest.mark.parametrize("model", [m for m in util.MODELS if m != "CAMF_C"])
@pytest.mark.parametrize("k", [10, 64, 128, 256])
def test_level_f32_rmse_within_1e5(model, k):
    """The throughput configuration: fp32 state, DPP-reduced dot, dependency-level schedule."""
    data = util.small_data(n_users=3000, n_items=400, n_dims=4, conds_per_dim=4, n=60000, seed=24)
    train, test = synth.split(data, 0.2)
    orc, inst = make_pair(model, train, k, 0)
    iters = 20
    o_losses, o_lrs, _ = orc.build_model(iters, util.LR, bold_driver=True)
    g_losses, g_lrs = inst.train(iters, util.LR, bold_driver=True)
    assert g_lrs.tolist() == o_lrs.tolist()             # same bold-driver decisions
    np.testing.assert_allclose(g_losses, o_losses, rtol=2e-5)
    tu, tj, tctx, tr = (test.u, test.j, test.ctx, test.r)
    if model in util.TWO_D:
        tctx = None
    oe = orc.eval_ratings(tu, tj, tctx, tr, 1.0, 5.0)
    ge = inst.eval_ratings(tu, tj, tctx, tr, 1.0, 5.0)
    assert abs(oe["RMSE"] - ge["RMSE"]) <= 1e-5          # north_star tolerance, fp32
    assert abs(oe["MAE"] - ge["MAE"]) <= 1e-5
    ot = orc.eval_ratings(*util.tuples_for(model, train), 1.0, 5.0)
    gt = inst.eval_ratings(*util.tuples_for(model, train), 1.0, 5.0) if model not in util.TWO_D else \
        inst.eval_ratings(*util.tuples_for(model, train)[:2], None, util.tuples_for(model, train)[3], 1.0, 5.0)
    assert abs(ot["RMSE"] - gt["RMSE"]) <= 1e-5


@pytest.mark.parametrize("model", ["CAMF_CI", "CAMF_CUCI", "BiasedMF"])
@pytest.mark.parametrize("k,n_dims", [(1, 1), (3, 2), (10, 4), (16, 4), (17, 3), (20, 6), (32, 8), (33, 2), (50, 5), (63, 12), (10, 16),
                                      (68, 3), (100, 4), (124, 2), (132, 4), (188, 3), (192, 2), (200, 4), (252, 5)])
def test_small_k_path_f32(model, k, n_dims):
    """k < 64 (the reference's default is 10): the 4 / 8 / 16-lanes-per-tuple kernels, every lane-count variant and
    ragged k; 64 < k < 256 with k % 4 == 0: the float4 kernel with masked slots.  Same bars as the k = 64/128/256 path: bold-driver decisions identical, loss 2e-5, RMSE/MAE 1e-5."""
    data = util.small_data(n_users=1500, n_items=300, n_dims=n_dims, conds_per_dim=3, n=30000, seed=27)
    train, test = synth.split(data, 0.2)
    orc, inst = make_pair(model, train, k, 0)
    assert inst.schedule_info()["levels"] >= 1
    o_losses, o_lrs, _ = orc.build_model(12, util.LR, bold_driver=True)
    g_losses, g_lrs = inst.train(12, util.LR, bold_driver=True)
    assert g_lrs.tolist() == o_lrs.tolist()
    np.testing.assert_allclose(g_losses, o_losses, rtol=2e-5)
    tctx = None if model in util.TWO_D else test.ctx
    oe = orc.eval_ratings(test.u, test.j, tctx, test.r, 1.0, 5.0)
    ge = inst.eval_ratings(test.u, test.j, tctx, test.r, 1.0, 5.0)
    assert abs(oe["RMSE"] - ge["RMSE"]) <= 1e-5 and abs(oe["MAE"] - ge["MAE"]) <= 1e-5
    for name, a in inst.get_states().items():
        ref = orc.state[name].reshape(a.shape)
        assert np.max(np.abs(ref - a)) <= 2e-4, name


@pytest.mark.parametrize("model,k,flags", [("CAMF_CI", 8, F64 | STRICT), ("CAMF_CUCI", 64, F64 | STRICT), ("BiasedMF", 10, F64 | STRICT),
                                           ("CAMF_CI", 128, 0), ("CAMF_CU", 10, 0), ("PMF", 70, 0)])
def test_heavy_tailed_items_use_the_tail_launch(model, k, flags):
    """Zipf item popularity: the hot items' chains give thousands of levels with a handful of tuples each; they are
    walked by one single-workgroup launch.  Strict fp64 stays bit-identical to the oracle, fp32 within the usual bars."""
    data = util.small_data(n_users=2500, n_items=300, n_dims=3, conds_per_dim=3, n=30000, seed=31, item_zipf=1.3)
    train, test = synth.split(data, 0.2)
    u, j, _, _ = util.tuples_for(model, train)
    _, off = capi.level_schedule(u, j, train.n_users, train.n_items)
    n_levels = len(off) - 1
    orc, inst = make_pair(model, train, k, flags)
    launches = inst.schedule_info()["levels"]
    assert n_levels > 1000 and launches < n_levels // 4, (n_levels, launches)
    for _ in range(3):
        lo, lg = orc.epoch(util.LR), inst.train_epoch(util.LR)
        assert abs(lo - lg) <= (1e-12 if flags else 2e-5) * abs(lo)
    if flags:
        assert_state_equal(orc, inst, exact=True)
    else:
        tctx = None if model in util.TWO_D else test.ctx
        oe = orc.eval_ratings(test.u, test.j, tctx, test.r, 1.0, 5.0)
        ge = inst.eval_ratings(test.u, test.j, tctx, test.r, 1.0, 5.0)
        assert abs(oe["RMSE"] - ge["RMSE"]) <= 1e-5 and abs(oe["MAE"] - ge["MAE"]) <= 1e-5


def test_camf_c_serial_f32():
    """CAMF_C (config C2 shape: k=64, fp32): condBias is shared by every tuple, so only the serial
    schedule is order-exact."""
    data = util.small_data(n_users=500, n_items=200, n_dims=4, conds_per_dim=3, n=12000, seed=25)
    train, test = synth.split(data, 0.2)
    orc, inst = make_pair("CAMF_C", train, 64, SERIAL)
    o_losses, o_lrs, _ = orc.build_model(15, util.LR, bold_driver=True)
    g_losses, g_lrs = inst.train(15, util.LR, bold_driver=True)
    assert g_lrs.tolist() == o_lrs.tolist()
    oe = orc.eval_ratings(test.u, test.j, test.ctx, test.r, 1.0, 5.0)
    ge = inst.eval_ratings(test.u, test.j, test.ctx, test.r, 1.0, 5.0)
    assert abs(oe["RMSE"] - ge["RMSE"]) <= 1e-5 and abs(oe["MAE"] - ge["MAE"]) <= 1e-5


@pytest.mark.parametrize("k,n_dims,flags", [(3, 1, 0), (64, 4, 0), (130, 8, 0), (256, 16, 0), (10, 3, F64), (100, 5, F64)])
def test_camf_c_conflict_free_blocks(k, n_dims, flags):
    """CAMF_C through sgd_camfc_blocks (parallel gather/dot/update inside runs of CRS tuples sharing no user and no item,
    sequential scalar condBias chain): bold-driver decisions and loss trajectory of the sequential oracle, RMSE/MAE
    within 1e-5 (fp32 state) / 1e-9 (fp64 state); CMI_NO_CAMFC_BLOCKS (the serial wave) must agree too."""
    data = util.small_data(n_users=900, n_items=700, n_dims=n_dims, conds_per_dim=3, n=9000, seed=33)
    train, test = synth.split(data, 0.2)
    orc, inst = make_pair("CAMF_C", train, k, SERIAL | flags)
    info = inst.schedule_info()
    assert info["kind"] == "serial" and info["flow_blocks"] > 0 and train.n / info["flow_blocks"] >= 3
    o_losses, o_lrs, _ = orc.build_model(10, util.LR, bold_driver=True)
    g_losses, g_lrs = inst.train(10, util.LR, bold_driver=True)
    assert g_lrs.tolist() == o_lrs.tolist()
    np.testing.assert_allclose(g_losses, o_losses, rtol=1e-10 if flags else 2e-5)
    tol = 1e-9 if flags else 1e-5
    oe = orc.eval_ratings(test.u, test.j, test.ctx, test.r, 1.0, 5.0)
    ge = inst.eval_ratings(test.u, test.j, test.ctx, test.r, 1.0, 5.0)
    assert abs(oe["RMSE"] - ge["RMSE"]) <= tol and abs(oe["MAE"] - ge["MAE"]) <= tol
    assert_state_equal(orc, inst, exact=False, atol=1e-9 if flags else 2e-4)


def test_camf_c_user_sorted_input_keeps_the_serial_wave():
    # consecutive CRS tuples of one user: every conflict-free run has length 1 -> no blocks
    data = util.small_data(n_users=40, n_items=300, n_dims=2, conds_per_dim=3, n=3000, seed=34)
    order = np.lexsort((data.j, data.u))
    import dataclasses
    srt = dataclasses.replace(data, u=data.u[order], j=data.j[order], ctx=data.ctx[order], r=data.r[order])
    orc, inst = make_pair("CAMF_C", srt, 16, SERIAL)
    assert inst.schedule_info()["flow_blocks"] == 0
    for _ in range(2):
        lo, lg = orc.epoch(util.LR), inst.train_epoch(util.LR)
        assert abs(lo - lg) <= 2e-5 * abs(lo)


def test_predict_batch_and_bounds():
    data = util.small_data(n_users=50, n_items=20, n=500, seed=26)
    orc, inst = make_pair("CAMF_CUCI", data, 16, F64)
    orc.epoch(util.LR)
    inst.train_epoch(util.LR)
    want = np.array([orc.predict(int(u), int(j), int(c)) for u, j, c in zip(data.u, data.j, data.ctx)])
    got = inst.predict(data.u, data.j, data.ctx)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    got_b = inst.predict(data.u, data.j, data.ctx, bound=(2.0, 4.0))
    np.testing.assert_allclose(got_b, np.clip(want, 2.0, 4.0), rtol=0, atol=1e-12)


def test_ragged_and_empty_contexts():
    """Contexts with different numbers of active conditions, including none (an ill-formed binary file
    can produce them; getConditions would return that many ids)."""
    rng = np.random.default_rng(3)
    nu, ni, nc, n = 40, 15, 7, 700
    ctx_lists = [[], [0], [1, 4], [0, 2, 5], [3, 4, 5, 6], [6]]
    ctx_ptr = np.cumsum([0] + [len(c) for c in ctx_lists]).astype(np.int32)
    ctx_conds = np.array([c for cl in ctx_lists for c in cl], dtype=np.int32)
    data = synth.RatingData(nu, ni, nc, 4, rng.integers(0, nu, n).astype(np.int32),
                            rng.integers(0, ni, n).astype(np.int32),
                            rng.integers(0, len(ctx_lists), n).astype(np.int32),
                            rng.integers(1, 6, n).astype(np.float64), ctx_ptr, ctx_conds)
    for model in ("CAMF_CI", "CAMF_CU", "CAMF_CUCI"):
        for flags in (F64 | STRICT, F64 | SERIAL | STRICT):
            orc, inst = make_pair(model, data, 6, flags)
            for _ in range(3):
                orc.epoch(util.LR)
                inst.train_epoch(util.LR)
            assert_state_equal(orc, inst, exact=True)
    orc, inst = make_pair("CAMF_CI", data, 64, 0)        # fp32 fast path with padded condition lanes
    for _ in range(3):
        lo, lg = orc.epoch(util.LR), inst.train_epoch(util.LR)
        assert abs(lo - lg) <= 1e-5 * abs(lo)
    assert_state_equal(orc, inst, exact=False, atol=2e-5)


def test_edge_cases_and_errors():
    data = util.small_data(n_users=10, n_items=5, n=40, seed=27)
    # empty training set: loss 0, state untouched
    inst = capi.Instance("CAMF_CI", 8, data.n_users, data.n_items, data.n_conds, flags=F64)
    inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, 3.0)
    inst.set_ratings(data.u[:0], data.j[:0], data.ctx[:0], data.r[:0], data.ctx_ptr, data.ctx_conds)
    st = synth.init_state("CAMF_CI", data, 8)
    inst.set_states(st)
    assert inst.train_epoch(0.01) == 0.0
    assert np.array_equal(inst.get_state("P"), st["P"])
    # single tuple
    orc, one = make_pair("CAMF_CU", data.subset(np.array([0])), 8, F64 | STRICT)
    assert orc.epoch(util.LR) == one.train_epoch(util.LR)
    assert_state_equal(orc, one, exact=True)
    # call-order and range errors are reported, not crashed on
    bad = capi.Instance("CAMF_CI", 8, data.n_users, data.n_items, data.n_conds)
    with pytest.raises(capi.CmiError):
        bad.train_epoch(0.01)                               # no ratings yet
    with pytest.raises(capi.CmiError):
        bad.set_ratings(data.u + 1000, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
    with pytest.raises(capi.CmiError):
        bad.set_state("itemBias", np.zeros(data.n_items))   # CAMF_CI has no itemBias
    with pytest.raises(capi.CmiError):
        bad.set_state("P", np.zeros(3))                     # wrong size
    with pytest.raises(capi.CmiError) as ei:
        capi.Instance("CAMF_C", 8, 4, 4, 4)                 # no exact parallel schedule for CAMF_C
    assert ei.value.code == capi.E_UNSUPPORTED


def test_nan_loss_is_an_error():
    data = util.small_data(n_users=30, n_items=10, n=400, seed=28)
    _, inst = make_pair("CAMF_CI", data, 8, F64)
    with pytest.raises(capi.CmiError) as ei:
        inst.train(50, 50.0, bold_driver=False)              # absurd learning rate diverges
    assert ei.value.code == capi.E_NUMERIC


def test_level_order_is_free():
    """Tuples inside a level commute: a different within-level order gives the bit-identical model."""
    import os
    data = util.small_data(n_users=800, n_items=90, n=9000, seed=29)
    outs = []
    for order in ("crs", "item", "user"):
        os.environ["CMI_LEVEL_ORDER"] = order
        try:
            _, inst = make_pair("CAMF_CI", data, 128, 0)
            for _ in range(3):
                inst.train_epoch(util.LR)
            outs.append(inst.get_states(np.float32))
        finally:
            os.environ.pop("CMI_LEVEL_ORDER", None)
    for other in outs[1:]:
        for name in outs[0]:
            assert np.array_equal(outs[0][name], other[name]), name


def test_dist_gpu_engine_aliases_device_state_and_exchange_is_identity_at_world1():
    """carskit_amd.dist on a real GPU: the exchange bucket of cmi_exchange_setup is aliased as a torch tensor (zero copy), the
    HIP pack kernel leaves this rank's item-side movement in it, and with one rank the RCCL reduce-scatter + all-gather exchange
    is the identity (to fp32 rounding)."""
    import os
    import socket
    import torch
    import torch.distributed as tdist
    from carskit_amd import dist as cdist
    data = util.small_data(n_users=400, n_items=50, n=5000, seed=33)
    _, a = make_pair("CAMF_CI", data, 128, 0)
    _, b = make_pair("CAMF_CI", data, 128, 0)
    eng = cdist.GpuEngine(a, 0)
    nq, nic = data.n_items * 128, data.n_items * data.n_conds
    assert eng.bucket.is_cuda and eng.bucket.numel() >= nq + nic and eng.bucket.numel() % 4 == 0
    q0, ic0 = a.get_state("Q", np.float32), a.get_state("icBias", np.float32)
    a.train_epoch(util.LR)
    b.train_epoch(util.LR)
    got = eng.pack()
    a.synchronize()
    got = got.cpu().numpy()
    assert np.array_equal(got[:nq].reshape(q0.shape), a.get_state("Q", np.float32) - q0)            # bucket = state - snapshot
    assert np.array_equal(got[nq:nq + nic].reshape(ic0.shape), a.get_state("icBias", np.float32) - ic0)
    eng.apply(1.0)                                         # snapshot + 1 * (state - snapshot): the same model to an fp32 ulp
    a.synchronize()
    np.testing.assert_allclose(a.get_state("Q", np.float32), b.get_state("Q", np.float32), rtol=1e-6, atol=1e-8)
    loss_t = eng.loss_tensor()
    assert loss_t.dtype == torch.float64 and abs(float(loss_t.item()) - a.last_loss()) == 0.0
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        runner = cdist.ShardedEpochRunner(a, tdist, device_index=0, always_exchange=True)
        assert runner.rs_ag                                # RCCL: in-place reduce-scatter + all-gather of the bucket
        assert runner.engine.lib_comm                      # ... issued by the LIBRARY (cmi_comm_*: the function cmi_group_* uses), not torch
        for _ in range(3):
            la, lb = runner.epoch(util.LR), b.train_epoch(util.LR)
            assert abs(la - lb) <= 1e-6 * abs(lb)
        # start + (x - start) re-rounds x in fp32, so the exchange is the identity only to an ulp
        for name in ("P", "Q", "userBias", "icBias"):
            np.testing.assert_allclose(a.get_state(name, np.float32), b.get_state(name, np.float32), rtol=1e-5, atol=1e-7)
    finally:
        tdist.destroy_process_group()


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

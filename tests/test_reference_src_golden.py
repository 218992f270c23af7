"""The oracle (and, on a GPU, the strict fp64 kernels) against the reference's OWN Java source, executed.

tests/golden/reference_src.json was minted in the build container by oracle/mint_reference_src.py: for BiasedMF, PMF, CAMF_C, CAMF_CI,
CAMF_CU and CAMF_CUCI the reference's `buildModel()` -- with its `predict`, `getConditions`, `isConverged`, `updateLRate` -- was run
statement by statement from the text of src/carskit/**.java by the Java-subset interpreter oracle/jvm/javasrc.py, on top of the vendored
librec jar's bytecode (oracle/jvm/interp.py).  The fixture holds inputs (rating cells, id maps, initial containers, hyper-parameters as
the Java fields hold them) and outputs (every epoch's loss and learning rate, the final containers), doubles as hex.

Bar: BIT-identical -- every loss, every bold-driver rate, every element of every container -- for the C oracle and the independent Python
restatement (CPU) and for the GPU's strict fp64 serial kernels (`-m gpu`).  This pins the statement order, the reads-before-writes, the
operator association and the CAMF_C loss quirk of the `src/carskit` loops against the reference's text itself rather than against a
reading of it."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_c
from tests import util

ALL = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_src.json")))["cases"]
SIM = ("SVD++", "CAMF_ICS", "CAMF_LCS", "CAMF_MCS")          # oracle/carskit_oracle_sim.c
GOLD = [c for c in ALL if c["model"] not in SIM]
GOLD_SIM = [c for c in ALL if c["model"] in SIM]
SHAPES = lambda c: {"P": (c["problem"]["n_users"], c["k"]), "Q": (c["problem"]["n_items"], c["k"]),
                    "userBias": (c["problem"]["n_users"],), "itemBias": (c["problem"]["n_items"],), "condBias": (c["problem"]["n_conds"],),
                    "ucBias": (c["problem"]["n_users"], c["problem"]["n_conds"]), "icBias": (c["problem"]["n_items"], c["problem"]["n_conds"]),
                    "Y": (c["problem"]["n_items"], c["k"]), "ccMatrix": (c["problem"]["n_conds"], c["problem"]["n_conds"]),
                    "cfMatrix": (c["problem"]["n_conds"], c.get("num_f", 0)), "cVector": (c["problem"]["n_conds"],)}


def fx(v):
    return float.fromhex(v)


def _inputs(c):
    p = c["problem"]
    cells = p["cells"]
    two_d = c["model"] in util.TWO_D or c["model"] == "SVD++"
    if two_d:   # the 2-D train matrix: mean over contexts per (user, item), CRS order (DataDAO.toTraditionalSparseMatrix)
        acc = {}
        for ui, _, v in cells:
            key = (p["ui_user"][ui], p["ui_item"][ui])
            s, n = acc.get(key, (0.0, 0))
            acc[key] = (s + v, n + 1)
        keys = sorted(acc)
        u = np.array([k[0] for k in keys], np.int32)
        j = np.array([k[1] for k in keys], np.int32)
        ctx = np.zeros(len(keys), np.int32)
        r = np.array([acc[k][0] / acc[k][1] for k in keys])
    else:
        u = np.array([p["ui_user"][ui] for ui, _, _ in cells], np.int32)
        j = np.array([p["ui_item"][ui] for ui, _, _ in cells], np.int32)
        ctx = np.array([cc for _, cc, _ in cells], np.int32)
        r = np.array([v for _, _, v in cells])
    conds = [[int(x) for x in key.split(",")] for key in p["ctx_keys"]]
    ctx_ptr = np.cumsum([0] + [len(x) for x in conds]).astype(np.int32)
    ctx_conds = np.array([x for cs in conds for x in cs], np.int32)
    shapes = SHAPES(c)
    state = {n: np.array([fx(x) for x in v]).reshape(shapes[n]) for n, v in c["init"].items()}
    return u, j, ctx, r, ctx_ptr, ctx_conds, state


def _test_tuples(c):
    p = c["problem"]
    t = c["test_cells"]
    return (np.array([p["ui_user"][ui] for ui, _, _ in t], np.int32), np.array([p["ui_item"][ui] for ui, _, _ in t], np.int32),
            np.array([cc for _, cc, _ in t], np.int32), np.array([v for _, _, v in t]))


def _check(c, losses, lrates, state):
    assert [float(x).hex() for x in losses] == c["epoch_loss"]
    assert [float(x).hex() for x in lrates] == c["epoch_lrate"]
    for n, want in c["final"].items():
        got = [float(x).hex() for x in np.asarray(state[n], dtype=np.float64).ravel()]
        assert got == want, n


@pytest.mark.parametrize("case", GOLD, ids=lambda c: "%s-k%d" % (c["model"], c["k"]))
def test_c_oracle_reproduces_the_interpreted_reference_source_bit_for_bit(case):
    u, j, ctx, r, ctx_ptr, ctx_conds, state = _inputs(case)
    p = case["problem"]
    gm = fx(case["global_mean"])
    orc = oracle_c.Oracle(case["model"], case["k"], p["n_users"], p["n_items"], p["n_conds"], u, j, ctx, r, ctx_ptr, ctx_conds, state, gm,
                          case["regU"], case["regI"], case["regB"], case["regC"])
    losses, lrates, _ = orc.build_model(case["iters"], case["lrate"], bold_driver=case["bold_driver"])
    _check(case, losses, lrates, orc.state)
    # Recommender.evalRatings (Recommender.java:504-594), executed from source over the held-out cells: MAE, RMSE, NMAE, rMAE, rRMSE
    tu, tj, tc, tr = _test_tuples(case)
    ev = orc.eval_ratings(tu, tj, tc, tr, 1.0, 5.0)
    for name in ("MAE", "RMSE", "NMAE", "rMAE", "rRMSE"):
        assert float(ev[name]).hex() == case["eval_ratings"][name], name
    assert fx(case["eval_ratings"]["MPE"]) == 0.0
    # the global mean the hosts compute from the contextual train matrix
    assert oracle_c.global_mean(np.array([v for _, _, v in p["cells"]])) == gm


@pytest.mark.parametrize("case", GOLD, ids=lambda c: "%s-k%d" % (c["model"], c["k"]))
def test_python_restatement_reproduces_it_too(case):
    from oracle import oracle_np
    u, j, ctx, r, ctx_ptr, ctx_conds, state = _inputs(case)
    p = case["problem"]
    conds = [ctx_conds[ctx_ptr[x]:ctx_ptr[x + 1]].tolist() for x in range(len(ctx_ptr) - 1)]
    m = oracle_np.MODELS[case["model"]](case["k"], p["n_users"], p["n_items"], p["n_conds"], conds, fx(case["global_mean"]), case["regU"],
                                        case["regI"], case["regB"], case["regC"])
    for n, a in state.items():
        setattr(m, n, a.tolist())
    lr, last, losses, lrates = case["lrate"], 0.0, [], []
    for it in range(1, case["iters"] + 1):
        lrates.append(lr)
        loss = m.epoch(list(zip(u.tolist(), j.tolist(), ctx.tolist(), r.tolist())), lr)
        losses.append(loss)
        if it > 1:                                   # IterativeRecommender.updateLRate, bold driver
            lr = lr * 1.05 if abs(last) > abs(loss) else lr * 0.5
        last = loss
    _check(case, losses, lrates, {n: np.array(getattr(m, n)) for n in case["final"]})


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLD, ids=lambda c: "%s-k%d" % (c["model"], c["k"]))
def test_gpu_strict_fp64_reproduces_the_interpreted_reference_source_bit_for_bit(case):
    """The PRODUCT path against the reference's own statements: STATE_F64 | SCHED_SERIAL | STRICT -- model, losses and bold-driver rates."""
    from carskit_amd import capi
    u, j, ctx, r, ctx_ptr, ctx_conds, state = _inputs(case)
    p = case["problem"]
    inst = capi.Instance(case["model"], case["k"], p["n_users"], p["n_items"], p["n_conds"],
                         flags=capi.FLAG_STATE_F64 | capi.FLAG_SCHED_SERIAL | capi.FLAG_STRICT)
    inst.set_hparams(case["regU"], case["regI"], case["regB"], case["regC"], fx(case["global_mean"]))
    if case["model"] in util.TWO_D:
        inst.set_ratings(u, j, None, r)
    else:
        inst.set_ratings(u, j, ctx, r, ctx_ptr, ctx_conds)
    inst.set_states(state)
    losses, lrates = inst.train(case["iters"], case["lrate"], bold_driver=case["bold_driver"])
    _check(case, losses, lrates, inst.get_states())
    tu, tj, tc, tr = _test_tuples(case)
    ev = inst.eval_ratings(tu, tj, None if case["model"] in util.TWO_D else tc, tr, 1.0, 5.0)
    for name in ("MAE", "RMSE", "NMAE", "rMAE", "rRMSE"):           # cmi_eval_ratings: tree-reduced dot and error sums, hence 1e-12
        assert abs(ev[name] - fx(case["eval_ratings"][name])) <= 1e-12, name
    # and the order-exact LEVEL schedule (the production schedule family): same model bits, loss to rounding
    if case["model"] != "CAMF_C":
        lv = capi.Instance(case["model"], case["k"], p["n_users"], p["n_items"], p["n_conds"], flags=capi.FLAG_STATE_F64 | capi.FLAG_STRICT)
        lv.set_hparams(case["regU"], case["regI"], case["regB"], case["regC"], fx(case["global_mean"]))
        if case["model"] in util.TWO_D:
            lv.set_ratings(u, j, None, r)
        else:
            lv.set_ratings(u, j, ctx, r, ctx_ptr, ctx_conds)
        lv.set_states(state)
        l2, r2 = lv.train(case["iters"], case["lrate"], bold_driver=case["bold_driver"])
        np.testing.assert_allclose(l2, [fx(x) for x in case["epoch_loss"]], rtol=1e-12)
        assert [float(x).hex() for x in r2] == case["epoch_lrate"]
        for n, want in case["final"].items():
            assert [float(x).hex() for x in lv.get_state(n).ravel()] == want, n


def _bold_loop(epoch, case):
    lr, last, losses, lrates = case["lrate"], 0.0, [], []
    for it in range(1, case["iters"] + 1):
        lrates.append(lr)
        loss = epoch(lr)
        losses.append(loss)
        if it > 1:
            lr = lr * 1.05 if abs(last) > abs(loss) else lr * 0.5
        last = loss
    return losses, lrates


@pytest.mark.parametrize("case", GOLD_SIM, ids=lambda c: "%s-k%d" % (c["model"], c["k"]))
def test_sim_oracle_reproduces_the_interpreted_reference_source_bit_for_bit(case):
    """SVD++ / CAMF_ICS / CAMF_LCS / CAMF_MCS (oracle/carskit_oracle_sim.c) against SVDPlusPlus.java / sim/CAMF_*.java, executed."""
    u, j, ctx, r, ctx_ptr, ctx_conds, state = _inputs(case)
    p = case["problem"]
    orc = oracle_c.SimOracle(case["model"], case["k"], p["n_users"], p["n_items"], p["n_conds"], u, j, ctx, r, ctx_ptr, ctx_conds,
                             np.array(case["empty_conds"], np.int32), state, fx(case["global_mean"]), case["regU"], case["regI"], case["regB"],
                             case["regC"], n_ctx_dims=case["n_ctx_dims"])
    losses, lrates = _bold_loop(orc.epoch, case)
    _check(case, losses, lrates, orc.state)


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLD_SIM, ids=lambda c: "%s-k%d" % (c["model"], c["k"]))
def test_gpu_strict_fp64_sim_models_reproduce_the_interpreted_reference_source(case):
    from carskit_amd import capi
    u, j, ctx, r, ctx_ptr, ctx_conds, state = _inputs(case)
    p = case["problem"]
    inst = capi.Instance(case["model"], case["k"], p["n_users"], p["n_items"], p["n_conds"],
                         flags=capi.FLAG_STATE_F64 | capi.FLAG_SCHED_SERIAL | capi.FLAG_STRICT)
    inst.set_hparams(case["regU"], case["regI"], case["regB"], case["regC"], fx(case["global_mean"]))
    if case["model"] != "SVD++":
        inst.set_sim_params(case["num_f"], case["n_ctx_dims"], case["empty_conds"])
        inst.set_ratings(u, j, ctx, r, ctx_ptr, ctx_conds)
    else:
        inst.set_ratings(u, j, None, r)
    inst.set_states(state)
    losses, lrates = inst.train(case["iters"], case["lrate"], bold_driver=case["bold_driver"])
    _check(case, losses, lrates, inst.get_states())


FM_GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_src.json")))["fm_cases"]


def _fm_inputs(c):
    p = c["problem"]
    u = np.array([p["ui_user"][ui] for ui, _, _ in p["cells"]], np.int32)
    j = np.array([p["ui_item"][ui] for ui, _, _ in p["cells"]], np.int32)
    ctx = np.array([cc for _, cc, _ in p["cells"]], np.int32)
    r = np.array([v for _, _, v in p["cells"]])
    pp = p["n_users"] + p["n_items"] + p["n_conds"]
    w = np.array([fx(x) for x in c["init"]["w"]])
    V = np.array([fx(x) for x in c["init"]["V"]]).reshape(pp, c["k"])
    return u, j, ctx, r, w, V


@pytest.mark.parametrize("case", FM_GOLD, ids=lambda c: "FM-k%d" % c["k"])
def test_fm_oracle_reproduces_the_interpreted_fm_source_bit_for_bit(case):
    """FM.java:115-220 (the dense ALS / coordinate-descent sweep, its pre-pass, its predict with the context-index quirk) executed from
    source against oracle/carskit_oracle_fm.c: w0, w, V after the sweeps and the predictions, bit for bit."""
    u, j, ctx, r, w, V = _fm_inputs(case)
    p = case["problem"]
    orc = oracle_c.FMOracle(case["k"], p["n_users"], p["n_items"], p["n_conds"], case["n_ctx_dims"], u, j, ctx, r, 0.0, w, V,
                            case["regLw"], case["regLf"])
    orc.init()
    loss = 0.0
    for _ in range(case["iters"]):
        loss = orc.sweep()
    assert float(orc.w0).hex() == case["final"]["w0"]
    assert [float(x).hex() for x in orc.w] == case["final"]["w"]
    assert [float(x).hex() for x in orc.V.ravel()] == case["final"]["V"]
    for uu, jj, cc, want in case["predictions"]:
        assert float(orc.predict(uu, jj, cc)).hex() == want
    assert float(loss).hex() == case["final_loss"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", FM_GOLD, ids=lambda c: "FM-k%d" % c["k"])
def test_gpu_fm_matches_the_interpreted_fm_source(case):
    """The GPU's sparse, exactly-equivalent formulation (tree-reduced sums, size * reg instead of one add per rating): 1e-8 relative on the
    model, 1e-9 on the predictions -- the bar of tests/test_gpu_fm.py, here against the reference's own statements."""
    from carskit_amd import capi
    u, j, ctx, r, w, V = _fm_inputs(case)
    p = case["problem"]
    g = capi.FMInstance(case["k"], p["n_users"], p["n_items"], p["n_conds"], case["n_ctx_dims"])
    g.set_hparams(case["regLw"], case["regLf"])
    g.set_ratings(u, j, ctx, r)
    g.set_model(0.0, w, V)
    g.init()
    for _ in range(case["iters"]):
        g.sweep()
    w0, gw, gV = g.get_model()
    assert abs(w0 - fx(case["final"]["w0"])) <= 1e-8 * max(1.0, abs(w0))
    np.testing.assert_allclose(gw, [fx(x) for x in case["final"]["w"]], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(gV.ravel(), [fx(x) for x in case["final"]["V"]], rtol=1e-6, atol=1e-9)

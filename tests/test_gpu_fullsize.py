"""BASELINE.json's configurations at their full (per-GPU) sizes on the GPU.

C3 (CAMF_CI k=128, 1 M users x 100 K items x 32 conditions, 50 M ratings): a direct one-epoch comparison with the CPU oracle
(about 30 s of single-thread CPU), plus size-independent properties -- idempotence at lr = 0, loss consistency with
evalRatings, and schedule independence (hub-chain levels, plain levels, eager launches and the two-lane graph give the
bit-identical model).
C5 (CAMF_CU k=256, one GPU's share of 10 M x 1 M x 128 conditions / 500 M ratings = 1.25 M users, 62.5 M ratings) and the
north_star shape (CAMF_CI k=128, 10 M x 1 M x 64 conditions, 200 M ratings): the same properties at full size, and a one-epoch
oracle comparison on a 5 M-tuple prefix (same id spaces, same tables).
C4 (FM k=64, one GPU's share of 5 M x 500 K x 64 / 200 M ratings = 625 K users, 25 M ratings): whole-train == init + sweeps,
phase-split == fused sweep, predictions == the FM formula, and the first seven phases of a sweep against a NumPy restatement of
the sparse formulation over all 25 M ratings."""
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
    # the CPU oracle's epoch over all 50 M tuples (about half a minute on one core) starts here, on a thread of its own, behind the
    # other C3 tests; test_c3_one_epoch_matches_oracle (the last of them) waits for it
    import threading
    orc = util.c_oracle("CAMF_CI", data, K, {n: a.astype(np.float64) for n, a in state.items()}, gm)
    _C3_ORACLE.clear()
    _C3_ORACLE["orc"] = orc
    _C3_ORACLE["thread"] = threading.Thread(target=lambda: _C3_ORACLE.__setitem__("loss", orc.epoch(util.LR)))
    _C3_ORACLE["thread"].start()
    return data, state, gm


_C3_ORACLE = {}


def _inst(c3, flags=0):
    data, state, gm = c3
    inst = capi.Instance("CAMF_CI", K, data.n_users, data.n_items, data.n_conds, flags=flags)
    inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
    inst.set_ratings(data.u, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
    inst.set_states(state)
    return inst


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
    for flags in (0, capi.FLAG_NO_CHAIN, capi.FLAG_NO_CHAIN | capi.FLAG_NO_GRAPH):
        inst = _inst(c3, flags)
        losses = [inst.train_epoch(util.LR) for _ in range(2)]
        outs.append((losses, inst.get_state("P", np.float32), inst.get_state("Q", np.float32),
                     inst.get_state("icBias", np.float32)))
        del inst
    for other in outs[1:]:
        np.testing.assert_allclose(other[0], outs[0][0], rtol=1e-12)
        for a, b in zip(outs[0][1:], other[1:]):
            assert np.array_equal(a, b)


def test_c3_one_epoch_matches_oracle(c3):
    data, state, gm = c3
    inst = _inst(c3)
    info = inst.schedule_info()
    assert info["tuples"] == data.n and info["kind"] == "chain-item" and 200 < info["levels"] < 400
    lg = inst.train_epoch(util.LR)
    _C3_ORACLE["thread"].join()
    orc, lo = _C3_ORACLE["orc"], _C3_ORACLE["loss"]
    assert abs(lg - lo) <= 1e-6 * abs(lo)                       # 50 M fp32 updates vs fp64, same order
    for name in ("P", "Q", "userBias", "icBias"):
        d = np.abs(inst.get_state(name, np.float64) - orc.state[name].reshape(inst.state_shape(name)))
        assert d.max() <= 2e-5, (name, d.max())
    # RMSE over a 2 M-tuple sample of the training set, clamped like evalRatings: the fp32 bar of the north star
    idx = np.arange(0, data.n, 25)
    ge = inst.eval_ratings(data.u[idx], data.j[idx], data.ctx[idx], data.r[idx], 1.0, 5.0)
    oe = orc.eval_ratings(data.u[idx], data.j[idx], data.ctx[idx], data.r[idx], 1.0, 5.0)
    assert abs(ge["RMSE"] - oe["RMSE"]) <= 1e-5 and abs(ge["MAE"] - oe["MAE"]) <= 1e-5


# ---------------------------------------------------------------------------------------------------------------------
# C5 share and the north_star shape
# ---------------------------------------------------------------------------------------------------------------------

BIG = {
    # name: (model, k, users, items, dims, conds/dim, ratings)
    "c5": ("CAMF_CU", 256, 1_250_000, 1_000_000, 4, 32, 62_500_000),
    "northstar": ("CAMF_CI", 128, 10_000_000, 1_000_000, 4, 16, 200_000_000),
}


def _big_inst(model, k, data, state, gm, flags=0, n=None):
    n = data.n if n is None else n
    inst = capi.Instance(model, k, data.n_users, data.n_items, data.n_conds, flags=flags)
    inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
    inst.set_ratings(data.u[:n], data.j[:n], data.ctx[:n], data.r[:n], data.ctx_ptr, data.ctx_conds)
    inst.set_states(state)
    return inst


def _reg_terms(model, data, state):
    """Regulariser part of the epoch loss at the given model: sum over tuples of regU|P[u]|^2 + regI|Q[j]|^2 + the bias terms.
    Per-row squared norms first (ranges of rows), then the per-tuple gathers in ranges of tuples on a few host threads (fp64 partial
    sums per range, added in range order)."""
    P, Q = state["P"], state["Q"]
    np2, nq2 = np.empty(len(P)), np.empty(len(Q))

    def norms(dst, M):
        def f(lo, hi):
            x = M[lo:hi].astype(np.float64)
            dst[lo:hi] = (x * x).sum(axis=1)
        synth._ranges(len(M), 1 << 18, f)

    norms(np2, P)
    norms(nq2, Q)
    ctab = data.ctx_conds.reshape(-1, data.n_dims)
    ci = model == "CAMF_CI"
    bias2 = (state["userBias"] if ci else state["itemBias"]).astype(np.float64) ** 2     # keyed by user (CAMF_CI) / item (CAMF_CU)
    cb2 = (state["icBias"] if ci else state["ucBias"]).astype(np.float64) ** 2           # [item or user][condition]
    step = 1 << 22
    parts = np.zeros((data.n + step - 1) // step)

    def tuples(lo, hi):
        u, j = data.u[lo:hi], data.j[lo:hi]
        conds = ctab[data.ctx[lo:hi]]
        row = j if ci else u
        acc = util.REG * (np2[u].sum() + nq2[j].sum() + bias2[u if ci else j].sum())
        for d in range(data.n_dims):
            acc += util.REGC * cb2[row, conds[:, d]].sum()
        parts[lo // step] = acc

    synth._ranges(data.n, step, tuples)
    return float(parts.sum())


@pytest.mark.parametrize("name", ["c5", "northstar"])
def test_big_shapes_full_size_properties_and_prefix_oracle(name):
    import os, sys, time
    T = [time.time()]
    def lap(w):
        if os.environ.get("CMI_TEST_TIMES"):
            print("LAP %s %s %.1f" % (name, w, time.time() - T[0]), file=sys.stderr, flush=True)
        T[0] = time.time()
    model, k, nu, ni, nd, cpd, nr = BIG[name]
    data = synth.generate_fast(nu, ni, nd, cpd, nr)
    lap("generate")
    state = synth.init_state(model, data, k, dtype=np.float32)
    gm = float(data.r.sum() / np.count_nonzero(data.r))
    lap("init_state")
    # the CPU oracle's epoch over the 5 M-tuple prefix (step 3) takes seconds on one core: it runs on a thread of its own behind steps 1-2.
    # Its arithmetic does not depend on the id values, so it runs on the users / items the prefix touches (compacted ids).
    import threading
    m = 5_000_000
    uu, ui = np.unique(data.u[:m], return_inverse=True)
    jj, ji = np.unique(data.j[:m], return_inverse=True)
    sub = synth.RatingData(len(uu), len(jj), data.n_conds, data.n_dims, ui.astype(np.int32), ji.astype(np.int32), data.ctx[:m],
                           data.r[:m], data.ctx_ptr, data.ctx_conds)
    rows = {"P": uu, "userBias": uu, "ucBias": uu, "Q": jj, "itemBias": jj, "icBias": jj}
    orc = util.c_oracle(model, sub, k, {n_: a[rows[n_]].astype(np.float64) for n_, a in state.items()}, gm)
    orc_loss = []
    orc_thread = threading.Thread(target=lambda: orc_loss.append(orc.epoch(util.LR)))
    orc_thread.start()

    # (1) lr = 0: nothing moves, and the epoch loss equals 0.5 * (sum e^2 + regularisers) at the initial model
    inst = _big_inst(model, k, data, state, gm)
    lap("instance")
    info = inst.schedule_info()
    assert info["tuples"] == data.n and info["kind"].startswith("chain")
    loss0 = inst.train_epoch(0.0)
    lap("epoch0")
    for n_, a in state.items():
        assert np.array_equal(inst.get_state(n_, np.float32), a), n_
    lap("get_state")
    step = 1 << 24
    e2 = 0.0
    for b in range(0, data.n, step):
        sl = slice(b, min(data.n, b + step))
        e2 += float(np.sum((data.r[sl] - inst.predict(data.u[sl], data.j[sl], data.ctx[sl])) ** 2))
    lap("predict")
    assert abs(loss0 - 0.5 * (e2 + _reg_terms(model, data, state))) <= 2e-6 * loss0
    lap("reg_terms")

    # (2) schedule independence at full size: hub-chain levels == plain level launches, bit for bit.  (On the C5 share and, in
    # test_c3_schedules_agree_bit_for_bit, on C3; the north_star shape would spend a minute building a second 200 M-tuple schedule
    # for the same statement.)
    if name == "c5":
        l1 = inst.train_epoch(util.LR)
        got = {n_: inst.get_state(n_, np.float32) for n_ in state}
        del inst
        plain = _big_inst(model, k, data, state, gm, flags=capi.FLAG_NO_CHAIN)
        assert plain.schedule_info()["kind"] == "level"
        plain.train_epoch(0.0)
        l2 = plain.train_epoch(util.LR)
        assert abs(l1 - l2) <= 1e-12 * abs(l2)
        for n_ in state:
            assert np.array_equal(plain.get_state(n_, np.float32), got[n_]), n_
        del plain, got
    else:
        del inst

    lap("step2")
    # (3) one epoch over the first 5 M tuples (full-size tables on the GPU) against the CPU oracle (started above) -- same tuples, same rows.
    pre = _big_inst(model, k, data, state, gm, n=m)
    lg = pre.train_epoch(util.LR)
    orc_thread.join()
    lo = orc_loss[0]
    assert abs(lg - lo) <= 1e-6 * abs(lo)
    for n_ in state:
        got_rows = pre.get_state(n_, np.float32)[rows[n_]].astype(np.float64)
        d = np.abs(got_rows - orc.state[n_].reshape(got_rows.shape))
        assert d.max() <= 2e-5, (n_, d.max())
    idx = np.arange(0, m, 5)
    ge = pre.eval_ratings(data.u[idx], data.j[idx], data.ctx[idx], data.r[idx], 1.0, 5.0)
    oe = orc.eval_ratings(sub.u[idx], sub.j[idx], sub.ctx[idx], sub.r[idx], 1.0, 5.0)
    assert abs(ge["RMSE"] - oe["RMSE"]) <= 1e-5 and abs(ge["MAE"] - oe["MAE"]) <= 1e-5
    lap("prefix")


# ---------------------------------------------------------------------------------------------------------------------
# C4 share: FM k=64, 625 K users x 500 K items x 64 conditions, 25 M ratings (FM.java:115-220)
# ---------------------------------------------------------------------------------------------------------------------

def _fm_predict_np(w0, w, V, nu, ni, nc, n_dims, u, j, c):
    """FM.predict (FM.java:93-113) in its pairwise form for the <= 3 non-zero features of a rating."""
    xc = 1.0 / n_dims
    has = c < nc
    cc = np.where(has, c, 0)
    vu, vj, vc = V[u], V[nu + j], V[nu + ni + cc] * (xc * has)[:, None]
    lin = w0 + w[u] + w[nu + j] + w[nu + ni + cc] * xc * has
    s = vu + vj + vc
    return lin + 0.5 * ((s * s).sum(axis=1) - (vu * vu).sum(axis=1) - (vj * vj).sum(axis=1) - (vc * vc).sum(axis=1))


def test_c4_fm_share_full_size():
    k = 64
    data = synth.generate_fast(625_000, 500_000, 4, 16, 25_000_000)
    p = data.n_users + data.n_items + data.n_conds
    rng = np.random.default_rng(1)
    w_init, V_init = rng.random(p), 0.1 * rng.standard_normal((p, k))
    regw, regf = synth.java_float(0.01), synth.java_float(0.02)

    def make():
        g = capi.FMInstance(k, data.n_users, data.n_items, data.n_conds, data.n_dims)
        g.set_hparams(regw, regf)
        g.set_ratings(data.u, data.j, data.ctx, data.r)
        g.set_model(0.0, w_init, V_init)
        return g

    a, b = make(), make()
    a.train(2)                                   # whole buildModel(): init + 2 sweeps
    b.init()
    b.sweep()                                    # fused sweep ...
    for ph in range(b.num_phases()):             # ... then one sweep as split reduce / apply phases (the multi-GPU form)
        b.phase_reduce(ph)
        b.phase_apply(ph)
    b.synchronize()
    ma, mb = a.get_model(), b.get_model()
    # the default form adds a coordinate's sums in an order the layout alone decides: cmi_fm_train == init + fused sweep + split phases,
    # bit for bit, at full size too (round 6; the relaxed LDS-atomic form is held to 1e-11 of it in tests/test_gpu_fm.py)
    assert ma[0] == mb[0]
    assert np.array_equal(ma[1], mb[1]) and np.array_equal(ma[2], mb[2])
    del b

    # predictions on a sample == the FM formula over the returned model
    idx = np.arange(0, data.n, 100)
    got = a.predict(data.u[idx], data.j[idx], data.ctx[idx])
    want = _fm_predict_np(ma[0], ma[1], ma[2], data.n_users, data.n_items, data.n_conds, data.n_dims,
                          data.u[idx].astype(np.int64), data.j[idx].astype(np.int64), data.ctx[idx].astype(np.int64))
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)

    # (the reference's w0 step ADDS (w0' - w0) to its error cache although the prediction moved the other way, FM.java:153-169:
    #  its incremental errors[] are by design not the residuals of the model, so "re-initialise and continue" is NOT an invariant)

    # the first seven phases of a sweep (w0; w of the user / item / context fields; column 0 of V for the three fields) against
    # a NumPy restatement of the same sparse field-parallel formulation over all 25 M ratings (np.bincount segmented sums)
    nu, ni, nc = data.n_users, data.n_items, data.n_conds
    u, j, c = data.u.astype(np.int64), data.j.astype(np.int64), data.ctx.astype(np.int64)
    has = c < nc
    cc = np.where(has, c, 0)
    xc = 1.0 / data.n_dims
    w0, w, V0 = 0.0, w_init.copy(), V_init[:, 0].copy()
    err = np.empty(data.n)
    step = 1 << 22
    for b0 in range(0, data.n, step):
        sl = slice(b0, min(data.n, b0 + step))
        err[sl] = data.r[sl] - _fm_predict_np(0.0, w_init, V_init, nu, ni, nc, data.n_dims, u[sl], j[sl], c[sl])
    size = float(data.n)
    upd = 0.0 - (err - w0).sum() / float(np.float32(size) + np.float32(regw))   # int + float: a float sum (FM.java:161)
    err += upd - w0
    w0 = upd
    fields = ((u, np.ones(data.n), None, 0, nu), (j, np.ones(data.n), None, nu, ni), (cc, np.full(data.n, xc), has, nu + ni, nc))
    q0 = V0[u] + V0[nu + j] + V0[nu + ni + cc] * (xc * has)
    for col, reg, is_w in ((w, regw, True), (V0, regf, False)):
        for idx, x, m, base, cnt in fields:
            theta = col[base + idx]
            h = x if is_w else x * q0 - x * x * theta
            num = (err - theta * h) * h
            den = h * h
            if m is not None:
                num, den = num * m, den * m
            newv = 0.0 - np.bincount(idx, weights=num, minlength=cnt) / (np.bincount(idx, weights=den, minlength=cnt) + size * reg)
            delta = (newv - col[base:base + cnt])[idx] * x
            if m is not None:
                delta = delta * m
            err += delta
            if not is_w:
                q0 += delta
            col[base:base + cnt] = newv
    g = make()
    g.init()
    for ph in range(7):
        g.phase_reduce(ph)
        g.phase_apply(ph)
    g.synchronize()
    gw0, gw, gV = g.get_model()
    assert abs(gw0 - w0) <= 1e-9 * max(1.0, abs(w0))
    np.testing.assert_allclose(gw, w, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(gV[:, 0], V0, rtol=1e-7, atol=1e-10)
    assert np.array_equal(gV[:, 1:], V_init[:, 1:])          # the other columns are untouched by these phases


def test_heavy_tailed_10m_ratings_owner_epoch_bit_identical_to_the_sequential_oracle():
    """SURVEY 8(d)'s heavy-tail stress at a size where every owner of the chip is busy: 10 M ratings, Zipf(1.1) items (the hottest item
    holds about 1.3 M of them = one dependency chain), CAMF_CI k=64.  The default schedule is the owner epoch (one persistent launch,
    user rows handed between owners as tagged records, millions of cross-XCD hand-offs per epoch).  In strict fp64 the model after
    three epochs is BIT-IDENTICAL to the sequential CPU oracle's; in fp32 it meets the north_star bar."""
    from oracle import oracle_c
    data = synth.generate(200_000, 20_000, 4, 8, 10_000_000, seed=77, item_zipf=1.1)
    k = 64
    state = synth.init_state("CAMF_CI", data, k)
    gm = oracle_c.global_mean(data.r)
    orc = util.c_oracle("CAMF_CI", data, k, state, gm)
    insts = []
    for flags in (capi.FLAG_STATE_F64 | capi.FLAG_STRICT, 0):
        inst = capi.Instance("CAMF_CI", k, data.n_users, data.n_items, data.n_conds, flags=flags)
        inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
        inst.set_ratings(data.u, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
        inst.set_states(state)
        assert inst.schedule_info()["kind"] == "owner-item"
        insts.append(inst)
    for _ in range(3):
        lo = orc.epoch(util.LR)
        l64, l32 = insts[0].train_epoch(util.LR), insts[1].train_epoch(util.LR)
        assert abs(lo - l64) <= 1e-10 * abs(lo)     # the same 10 M terms in another association (per-owner, per-lane partial sums)
        assert abs(lo - l32) <= 3e-5 * abs(lo)
    for name, a in insts[0].get_states().items():
        assert np.array_equal(orc.state[name].reshape(a.shape), a), name
    for name, a in insts[1].get_states().items():
        assert np.max(np.abs(orc.state[name].reshape(a.shape) - a)) <= 1e-3, name   # fp32 along a 1.3 M-step recurrence

"""The hub-chain level kernel (carskit_amd/csrc/chain_kernels.hip; the default path on wide data) on the GPU.

Bars:
  * fp32 state: model state BIT-IDENTICAL to the plain level schedule's (same per-tuple expressions, and the chain schedule
    commutes exactly -- tests/test_chain_schedule.py), loss to 1e-12 relative (same terms, other summation tree); and the
    north_star bar against the oracle (RMSE/MAE within 1e-5);
  * fp64 state (the fp64 fast path): RMSE/MAE within 1e-9 of the oracle, state within 1e-12.
"""
import os

import numpy as np
import pytest

from carskit_amd import capi, synth
from oracle import oracle_c
from tests import util
from tests.test_gpu_parity import assert_state_equal, make_pair

pytestmark = pytest.mark.gpu

CHAIN, NOCHAIN, F64 = capi.FLAG_SCHED_CHAIN, capi.FLAG_NO_CHAIN, capi.FLAG_STATE_F64
LEVEL_MODELS = [m for m in util.MODELS if m != "CAMF_C"]


def _with_hub(hub, fn):
    old = os.environ.get("CMI_CHAIN_HUB")
    os.environ["CMI_CHAIN_HUB"] = hub
    try:
        return fn()
    finally:
        if old is None:
            del os.environ["CMI_CHAIN_HUB"]
        else:
            os.environ["CMI_CHAIN_HUB"] = old


@pytest.mark.parametrize("model", LEVEL_MODELS)
@pytest.mark.parametrize("k", [10, 31, 48, 64, 100, 128, 256])
@pytest.mark.parametrize("hub", ["item", "user"])
def test_chain_f32_state_bitwise_equals_plain_levels(model, k, hub):
    data = util.small_data(n_users=3000, n_items=300, n_dims=4, conds_per_dim=4, n=50000, seed=31)
    _, plain = make_pair(model, data, k, NOCHAIN)
    _, chain = _with_hub(hub, lambda: make_pair(model, data, k, CHAIN))
    info = chain.schedule_info()
    assert info["kind"] == "chain-" + hub
    u, j, _, _ = util.tuples_for(model, data)
    assert info["levels"] < len(capi.level_schedule(u, j, data.n_users, data.n_items)[1]) - 1
    lr = util.LR
    for _ in range(3):
        lp, lc = plain.train_epoch(lr), chain.train_epoch(lr)
        assert abs(lp - lc) <= 1e-12 * abs(lp)
    sp, sc = plain.get_states(), chain.get_states()
    for name in sp:
        assert np.array_equal(sp[name], sc[name]), name


@pytest.mark.parametrize("model", LEVEL_MODELS)
@pytest.mark.parametrize("hub", ["item", "user"])
def test_chain_f32_vs_oracle_north_star_bar(model, hub):
    data = util.small_data(n_users=3000, n_items=400, n_dims=4, conds_per_dim=4, n=60000, seed=24)
    train, test = synth.split(data, 0.2)
    orc, inst = _with_hub(hub, lambda: make_pair(model, train, 128, CHAIN))
    o_losses, o_lrs, _ = orc.build_model(20, util.LR, bold_driver=True)
    g_losses, g_lrs = inst.train(20, util.LR, bold_driver=True)
    assert g_lrs.tolist() == o_lrs.tolist()
    np.testing.assert_allclose(g_losses, o_losses, rtol=2e-5)
    tctx = None if model in util.TWO_D else test.ctx
    oe = orc.eval_ratings(test.u, test.j, tctx, test.r, 1.0, 5.0)
    ge = inst.eval_ratings(test.u, test.j, tctx, test.r, 1.0, 5.0)
    assert abs(oe["RMSE"] - ge["RMSE"]) <= 1e-5 and abs(oe["MAE"] - ge["MAE"]) <= 1e-5


@pytest.mark.parametrize("model", LEVEL_MODELS)
@pytest.mark.parametrize("k", [32, 64, 70, 128, 256])
@pytest.mark.parametrize("hub", ["item", "user"])
def test_chain_f64_fast_path_vs_oracle(model, k, hub):
    data = util.small_data(n_users=400, n_items=60, n_dims=3, conds_per_dim=4, n=8000, seed=23)
    train, test = synth.split(data, 0.2)
    orc, inst = _with_hub(hub, lambda: make_pair(model, train, k, F64 | CHAIN))
    assert inst.schedule_info()["kind"] == "chain-" + hub
    o_losses, _, _ = orc.build_model(10, util.LR, bold_driver=True)
    g_losses, _ = inst.train(10, util.LR, bold_driver=True)
    np.testing.assert_allclose(g_losses, o_losses, rtol=1e-11)
    tctx = None if model in util.TWO_D else test.ctx
    oe = orc.eval_ratings(test.u, test.j, tctx, test.r, 1.0, 5.0)
    ge = inst.eval_ratings(test.u, test.j, tctx, test.r, 1.0, 5.0)
    for key in ("MAE", "RMSE", "NMAE", "rMAE", "rRMSE"):
        assert abs(oe[key] - ge[key]) <= 1e-9, key
    assert_state_equal(orc, inst, exact=False, atol=1e-12)


@pytest.mark.parametrize("model", LEVEL_MODELS)
@pytest.mark.parametrize("k,n_dims", [(1, 1), (5, 3), (10, 2), (16, 4), (17, 4), (20, 7), (32, 8), (33, 2), (63, 12)])
def test_chain_small_k_lane_layouts_bitwise_equal_plain(model, k, n_dims):
    """sgd_chain_small (fp32, k < 64; the reference's default num.factors is 10): 4, 8 or 16 lanes per unit by (k, D), ragged rows,
    more condition ids per unit than 4 per lane; vs the plain level schedule's small-k kernel, and vs the oracle at the north_star bar."""
    data = synth.generate(2000, 150, n_dims, 3, 30000, seed=100 + k)
    for hub in ("item", "user"):
        _, plain = make_pair(model, data, k, NOCHAIN)
        orc, chain = _with_hub(hub, lambda: make_pair(model, data, k, CHAIN))
        assert chain.schedule_info()["kind"] == "chain-" + hub
        for _ in range(3):
            lp, lc, lo = plain.train_epoch(util.LR), chain.train_epoch(util.LR), orc.epoch(util.LR)
            assert abs(lp - lc) <= 1e-12 * abs(lp)
            assert abs(lo - lc) <= 3e-5 * abs(lo)
        sp, sc = plain.get_states(), chain.get_states()
        for name in sp:
            assert np.array_equal(sp[name], sc[name]), (hub, name)
        assert_state_equal(orc, chain, exact=False, atol=3e-4)


@pytest.mark.parametrize("model", ["CAMF_CI", "CAMF_CU", "CAMF_CUCI"])
def test_chain_many_conditions_and_dimensions(model):
    """> 64 conditions (the LDS row is filled / written back by the remainder loops) and 6 dimensions with long units
    (more than 64 condition ids staged per unit)."""
    data = synth.generate(1500, 40, 6, 14, 40000, seed=41)   # 84 conditions, D = 6, ~1000 ratings per item
    for hub in ("item", "user"):
        _, plain = make_pair(model, data, 64, NOCHAIN)
        _, chain = _with_hub(hub, lambda: make_pair(model, data, 64, CHAIN))
        for _ in range(2):
            lp, lc = plain.train_epoch(util.LR), chain.train_epoch(util.LR)
            assert abs(lp - lc) <= 1e-12 * abs(lp)
        sp, sc = plain.get_states(), chain.get_states()
        for name in sp:
            assert np.array_equal(sp[name], sc[name]), (hub, name)


def test_chain_repeated_pairs_in_many_contexts():
    """DePaulMovie-like data: the same (user, item) pair rated in several contexts = consecutive CRS tuples sharing BOTH rows;
    they must never share a unit."""
    data = synth.generate(40, 30, 2, 3, 6000, seed=43)      # 1200 pairs x up to 9 contexts
    for model in ("CAMF_CI", "CAMF_CUCI"):
        orc, inst = make_pair(model, data, 64, F64 | CHAIN)
        for _ in range(3):
            lo, lg = orc.epoch(util.LR), inst.train_epoch(util.LR)
            assert abs(lo - lg) <= 1e-11 * abs(lo)
        assert_state_equal(orc, inst, exact=False, atol=1e-12)


def test_chain_is_the_default_on_wide_data_and_not_on_narrow():
    wide = synth.generate_fast(200_000, 20_000, 4, 8, 4_000_000, seed=3)
    state = synth.init_state("CAMF_CI", wide, 64, dtype=np.float32)
    inst = capi.Instance("CAMF_CI", 64, wide.n_users, wide.n_items, wide.n_conds)
    inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, 3.0)
    inst.set_ratings(wide.u, wide.j, wide.ctx, wide.r, wide.ctx_ptr, wide.ctx_conds)
    inst.set_states(state)
    assert inst.schedule_info()["kind"] == "chain-item"
    assert np.isfinite(inst.train_epoch(util.LR))
    narrow = util.small_data(n_users=300, n_items=40, n=4000, seed=22)
    _, inst2 = make_pair("CAMF_CI", narrow, 64, 0)
    assert inst2.schedule_info()["kind"] == "level"
    with pytest.raises(capi.CmiError):       # forcing it where no chain kernel exists is an error, not a silent fallback
        make_pair("CAMF_CI", narrow, 300, CHAIN)


def test_forced_chain_on_heavy_tailed_items_bitwise_equals_plain():
    """Heavy-tailed item degrees: the hot items' chains force tens of thousands of narrow levels.  Forced, the hub-chain schedule still
    runs (one launch per level since round 4: the multi-level workgroup walk sgd_chain_tail measured slower than the plain narrow-run
    walk and was removed) and is bit-identical to the plain level schedule.  It is NOT chosen automatically there: the automatic
    choice for such data is the owner epoch."""
    data = synth.generate(20000, 2000, 4, 4, 400_000, seed=51, item_zipf=0.9)
    for model, k in (("CAMF_CI", 128), ("CAMF_CU", 64), ("BiasedMF", 64)):
        _, plain = make_pair(model, data, k, NOCHAIN | capi.FLAG_NO_OWNER)
        _, auto = make_pair(model, data, k, 0)
        _, chain = make_pair(model, data, k, CHAIN)
        kind = chain.schedule_info()["kind"]
        assert auto.schedule_info()["kind"] == "owner-item" and kind.startswith("chain-")   # narrow levels: the owner epoch by default
        for _ in range(2):
            lp, lc = plain.train_epoch(util.LR), chain.train_epoch(util.LR)
            assert abs(lp - lc) <= 1e-12 * abs(lp)
        sp, sc = plain.get_states(), chain.get_states()
        for name in sp:
            assert np.array_equal(sp[name], sc[name]), (model, name)
    # fp64 state through the same launches, against the oracle
    orc, inst = make_pair("CAMF_CI", data, 64, F64 | CHAIN)
    for _ in range(2):
        lo_, lg = orc.epoch(util.LR), inst.train_epoch(util.LR)
        assert abs(lo_ - lg) <= 1e-11 * abs(lo_)
    assert_state_equal(orc, inst, exact=False, atol=1e-12)


ARENA = capi.FLAG_SPOKE_ARENA


@pytest.mark.parametrize("model", LEVEL_MODELS)
@pytest.mark.parametrize("k,flags", [(64, 0), (100, 0), (128, 0), (256, 0), (64, F64), (128, F64)])
@pytest.mark.parametrize("hub", ["item", "user"])
def test_spoke_arena_is_bit_identical_and_every_reader_sees_the_live_rows(model, k, flags, hub):
    """Round 3: the spoke arena (SgdArgs::arena; the default for spoke tables of 2 GiB and more, forced here) moves WHERE a spoke row
    lives between its tuples -- read from the slot of its own tuple, written to the slot of the row's next tuple -- not what is computed:
    the model must equal the table-resident form bit for bit, and everything that reads the model between epochs (get_state, evaluation,
    prediction, a rewritten container, save / load) must see the rows that currently live in the arena."""
    data = util.small_data(n_users=1200, n_items=260, n_dims=3, conds_per_dim=4, n=24000, seed=41)
    train, test = synth.split(data, 0.2)
    _, ref = _with_hub(hub, lambda: make_pair(model, train, k, CHAIN | flags | capi.FLAG_NO_ARENA))
    orc, arena = _with_hub(hub, lambda: make_pair(model, train, k, CHAIN | flags | ARENA))
    assert arena.schedule_traffic()["spoke_arena"] and not ref.schedule_traffic()["spoke_arena"]
    tctx = None if model in util.TWO_D else test.ctx
    for ep in range(4):
        lr_, la_ = ref.train_epoch(util.LR), arena.train_epoch(util.LR)
        assert lr_ == la_, (ep, lr_, la_)
        if ep == 1:      # readers in the middle of training
            for name, a in ref.get_states().items():
                assert np.array_equal(a, arena.get_state(name)), name
            er = ref.eval_ratings(test.u, test.j, tctx, test.r, 1.0, 5.0)
            ea = arena.eval_ratings(test.u, test.j, tctx, test.r, 1.0, 5.0)
            assert er == ea
            assert np.array_equal(ref.predict(test.u[:64], test.j[:64], None if tctx is None else tctx[:64]),
                                  arena.predict(test.u[:64], test.j[:64], None if tctx is None else tctx[:64]))
        if ep == 2:      # a container rewritten from the host between epochs (both the arena-backed one and the other side)
            st = ref.get_states()
            # (round 4: the arena slots also carry the spoke row's scalar bias -- userBias / itemBias are arena-backed containers too:
            # rewriting ONE of the two must not lose the other's live values)
            for name in ("userBias", "P", "itemBias", "Q"):
                if name not in st:
                    continue
                new = (st[name] * 0.5).astype(st[name].dtype)
                ref.set_state(name, new)
                arena.set_state(name, new)
                for other, a in ref.get_states().items():
                    assert np.array_equal(a, arena.get_state(other)), (name, other)
    for name, a in ref.get_states().items():
        assert np.array_equal(a, arena.get_state(name)), name


def test_spoke_arena_survives_save_load_and_a_new_rating_set(tmp_path):
    data = util.small_data(n_users=900, n_items=200, n_dims=3, conds_per_dim=3, n=15000, seed=43)
    _, ref = make_pair("CAMF_CI", data, 128, CHAIN | capi.FLAG_NO_ARENA)
    _, a = make_pair("CAMF_CI", data, 128, CHAIN | ARENA)
    for _ in range(2):
        ref.train_epoch(util.LR)
        a.train_epoch(util.LR)
    p = tmp_path / "arena.cmi"
    a.save_model(p)                                              # goes through get_state: the live rows
    _, b = make_pair("CAMF_CI", data, 128, CHAIN | ARENA, seed=9)
    b.load_model(p)                                              # goes through set_state: the arena is refilled before the next epoch
    ref.train_epoch(util.LR)
    b.train_epoch(util.LR)
    for name, x in ref.get_states().items():
        assert np.array_equal(x, b.get_state(name)), name
    # a second cmi_set_ratings on the same handle: the live rows come home before the arena is rebuilt
    half = data.subset(np.arange(data.n // 2))
    for inst in (ref, b):
        inst.set_ratings(half.u, half.j, half.ctx, half.r, half.ctx_ptr, half.ctx_conds)
        inst.train_epoch(util.LR)
    for name, x in ref.get_states().items():
        assert np.array_equal(x, b.get_state(name)), name


def test_spoke_arena_with_the_item_side_in_the_arena_through_the_exchange():
    """hub = user puts Q -- a container the multi-GPU exchange packs and rewrites -- into the arena: a group of two shards on one
    device must still equal the table-resident group bit for bit."""
    model, k = "CAMF_CU", 64
    data = util.small_data(n_users=700, n_items=160, n_dims=3, conds_per_dim=3, n=14000, seed=47)
    gm = oracle_c.global_mean(data.r)
    state = synth.init_state(model, data, k, seed=5, dtype=np.float32)

    def group(flags):
        g = capi.Group(model, k, data.n_users, data.n_items, data.n_conds, 2, devices=[0, 0], flags=CHAIN | flags)
        g.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
        g.set_ratings(data.u, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
        g.set_states(state)
        return g
    ga, gr = _with_hub("user", lambda: group(ARENA)), _with_hub("user", lambda: group(capi.FLAG_NO_ARENA))
    assert ga.member(0).schedule_info()["kind"] == "chain-user" and ga.member(0).schedule_traffic()["spoke_arena"]
    for _ in range(3):
        assert ga.train_epoch(util.LR) == gr.train_epoch(util.LR)
    for name, x in gr.get_states(np.float32).items():
        assert np.array_equal(x, ga.get_state(name, np.float32)), name


@pytest.mark.parametrize("hub,name", [("user", "Q"), ("item", "P")])
def test_spoke_arena_honours_host_writes_through_the_device_pointer(hub, name):
    """cmi_state_device_ptr is documented for the host's in-place epoch-boundary exchange, i.e. the host WRITES the table behind
    it.  With the container's live rows in the spoke arena such a write used to be lost (the next epoch read the arena, the following
    gather overwrote the table).  Writing through the pointer must equal the cmi_set_state path bit for bit."""
    import torch
    from carskit_amd.dist import _DevArray
    model, k = "CAMF_CU", 64
    data = util.small_data(n_users=900, n_items=220, n_dims=3, conds_per_dim=3, n=16000, seed=49)
    _, a = _with_hub(hub, lambda: make_pair(model, data, k, CHAIN | ARENA))
    _, b = _with_hub(hub, lambda: make_pair(model, data, k, CHAIN | ARENA))
    assert a.schedule_traffic()["spoke_arena"]
    for ep in range(3):
        la, lb = a.train_epoch(util.LR), b.train_epoch(util.LR)
        assert la == lb
        new = (a.get_state(name) * 0.75).astype(np.float32)
        a.set_state(name, new)                                          # the path that always worked
        ptr, cnt, dt = b.state_device_ptr(name)                         # the documented in-place path
        view = torch.as_tensor(_DevArray(ptr, cnt, dt), device=torch.device("cuda", 0))
        view.copy_(torch.from_numpy(new.reshape(-1)).to(view.device))
        torch.cuda.synchronize()
    a.train_epoch(util.LR)
    b.train_epoch(util.LR)
    for n_, x in a.get_states().items():
        assert np.array_equal(x, b.get_state(n_)), n_


@pytest.mark.parametrize("hub,extra,probe", [("item", 0, "1"), ("user", 0, "1"), ("item", capi.FLAG_NO_GRAPH, "-1"), ("item", 0, "-1")])
def test_arena_probe_times_both_forms_at_rate_zero_and_leaves_the_model_alone(hub, extra, probe):
    """Spoke tables of 256 MiB .. 2 GiB (BASELINE C5's share): table or arena is the BOX's choice, so the first training call times
    one epoch of each form at learning rate 0 -- x + 0 * (...) = x: nothing moves -- and keeps the faster (forced on this small set
    with CMI_ARENA_PROBE=1).  Whatever it picks, the model equals the table-resident run bit for bit, epoch by epoch, and the
    schedule note says what was measured.  probe "-1" forces the "table wins" verdict; with FLAG_NO_GRAPH there is then no captured graph
    to drop (the unguarded hipGraphExecDestroy(nullptr) left a sticky error for the first real epoch)."""
    import os
    model, k = "CAMF_CU", 64
    data = util.small_data(n_users=900, n_items=220, n_dims=3, conds_per_dim=3, n=16000, seed=53)
    _, ref = _with_hub(hub, lambda: make_pair(model, data, k, CHAIN | capi.FLAG_NO_ARENA | extra))
    os.environ["CMI_ARENA_PROBE"] = probe
    try:
        _, prb = _with_hub(hub, lambda: make_pair(model, data, k, CHAIN | extra))
    finally:
        del os.environ["CMI_ARENA_PROBE"]
    assert prb.schedule_traffic()["spoke_arena"]                 # built, choice pending
    before = prb.get_states()
    for ep in range(3):
        assert ref.train_epoch(util.LR) == prb.train_epoch(util.LR)
        if ep == 0:
            note = prb.schedule_note()
            assert "spoke arena probe: table " in note and (note.endswith("-> arena") or note.endswith("-> table"))
            assert prb.schedule_traffic()["spoke_arena"] == note.endswith("-> arena")
            assert probe != "-1" or note.endswith("-> table")
    for name, a in ref.get_states().items():
        assert np.array_equal(a, prb.get_state(name)), name
    assert any(not np.array_equal(before[n], prb.get_state(n)) for n in before)     # (and it did train)

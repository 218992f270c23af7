"""cmi_group_*: one recommender over several shards from ONE process (the entry the Java / C++ hosts use).

A single-GPU box runs a group of N shards on device 0 through the in-process exchange; it must equal, bit for bit, the same
algorithm driven by hand over N separate capi.Instances with the library's own pack / apply kernels and a float32 sum of the buckets
in shard order -- which is exactly what carskit_amd.dist.ShardedEpochRunner does with gloo's all-reduce at world size 2
(tests/test_gpu_dist_two_ranks.py ties that runner to the host-side simulation).  The RCCL path (one shard per DEVICE,
ncclCommInitAll refuses two ranks on one device) cannot run on a single-GPU box: it is the same pack / apply kernels around
ncclReduceScatter + ncclAllGather and is exercised by multi-GPU runs only -- said plainly in DESIGN.md section 7.  Against the oracle: a group of ONE is the plain
instance, i.e. the order-exact path; a group of N is N local order-exact passes merged, checked against N oracles merged the same
way on the host."""
import ctypes as C

import numpy as np
import pytest

from carskit_amd import capi, dist as cdist, synth
from tests import util

pytestmark = pytest.mark.gpu


def _problem(model, k, seed=5, n_users=600, n_items=150, n=12000):
    data = util.small_data(n_users=n_users, n_items=n_items, n_dims=3, conds_per_dim=3, n=n, seed=seed)
    train, test = synth.split(data, 0.2)
    from oracle import oracle_c
    gm = oracle_c.global_mean(train.r)
    state = synth.init_state(model, train, k, seed=seed + 1, dtype=np.float32)
    return train, test, gm, state


def _group(model, k, train, gm, state, shards, flags=0):
    g = capi.Group(model, k, train.n_users, train.n_items, train.n_conds, shards, devices=[0] * shards, flags=flags)
    g.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
    u, j, ctx, r = util.tuples_for(model, train)
    g.set_ratings(u, j, ctx, r, train.ctx_ptr, train.ctx_conds)
    g.set_states(state)
    return g


def _manual_shards(model, k, train, gm, state, world, flags=0):
    """The same shards as separate instances (cut by the Python rule the library restates)."""
    insts, cuts = [], []
    u, j, ctx, r = util.tuples_for(model, train)
    d2 = synth.RatingData(train.n_users, train.n_items, train.n_conds, train.n_dims, u, j, ctx, r, train.ctx_ptr, train.ctx_conds,
                          train.min_rate, train.max_rate, dict(train.meta))
    for rank in range(world):
        shard, (lo, hi) = cdist.shard_by_user(d2, rank, world)
        inst = capi.Instance(model, k, hi - lo, train.n_items, train.n_conds, flags=flags)
        inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
        inst.set_ratings(shard.u, shard.j, shard.ctx, shard.r, train.ctx_ptr, train.ctx_conds)
        inst.set_states({n: (a[lo:hi] if n in ("P", "userBias", "ucBias") else a) for n, a in state.items()})
        insts.append(inst)
        cuts.append((lo, hi))
    return insts, cuts


@pytest.mark.parametrize("model", ["CAMF_CI", "CAMF_CU", "CAMF_CUCI", "BiasedMF", "PMF"])
@pytest.mark.parametrize("world", [2, 3])
def test_group_on_one_device_equals_manual_exchange_bitwise(model, world):
    k = 64
    train, test, gm, state = _problem(model, k)
    g = _group(model, k, train, gm, state, world)
    assert g.size() == world
    insts, cuts = _manual_shards(model, k, train, gm, state, world)
    for s in range(world):
        si = g.shard_info(s)
        assert (si["user_lo"], si["user_hi"]) == cuts[s] and si["exchange"] == "in-process" and si["device"] == 0
    # manual exchange with the library's kernels: bucket_s = item_side_s - snapshot ; sum in shard order (fp32) ; apply 1/W
    buckets = []
    for inst in insts:
        ptr, cnt, dt = inst.exchange_setup(pad_to=world)
        buckets.append((ptr, cnt, dt))
    hip = C.CDLL("libamdhip64.so")
    for _ in range(3):
        lg = g.train_epoch(util.LR)
        local = [inst.train_epoch(util.LR) for inst in insts]
        host = []
        for inst, (ptr, cnt, dt) in zip(insts, buckets):
            inst.exchange_pack()
            inst.synchronize()
            b = np.empty(cnt, dtype=dt)
            assert hip.hipMemcpy(b.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(b.nbytes), 2) == 0
            host.append(b)
        total = host[0].copy()
        for b in host[1:]:
            total += b                                  # same element type, same order as the in-process exchange
        for inst, (ptr, cnt, dt) in zip(insts, buckets):
            assert hip.hipMemcpy(C.c_void_p(ptr), total.ctypes.data_as(C.c_void_p), C.c_size_t(total.nbytes), 1) == 0
            inst.exchange_apply(1.0 / world)
            inst.synchronize()
        tot = local[0]
        for x in local[1:]:
            tot += x
        assert lg == tot, (lg, tot)
    got = g.get_states(np.float32)
    for name, a in got.items():
        if name in ("P", "userBias", "ucBias"):
            ref = np.concatenate([inst.get_state(name, np.float32) for inst in insts])
        else:
            ref = insts[0].get_state(name, np.float32)
            for inst in insts[1:]:
                assert np.array_equal(ref, inst.get_state(name, np.float32)), name
        assert np.array_equal(a, ref), name
    # evaluation: routed to the owning shards and merged
    e = g.eval_ratings(test.u, test.j, test.ctx, test.r, 1.0, 5.0)
    sums = np.zeros(2)
    cnt = 0
    for inst, (lo, hi) in zip(insts, cuts):
        m = (test.u >= lo) & (test.u < hi)
        if m.any():
            ei = inst.eval_ratings(test.u[m] - lo, test.j[m], test.ctx[m], test.r[m], 1.0, 5.0)
            sums += [ei["MAE"] * ei["n"], ei["RMSE"] ** 2 * ei["n"]]
            cnt += ei["n"]
    assert e["n"] == cnt == test.n
    assert abs(e["MAE"] - sums[0] / cnt) < 1e-12 and abs(e["RMSE"] - np.sqrt(sums[1] / cnt)) < 1e-12
    p = g.predict_batch(test.u[:50], test.j[:50], test.ctx[:50], bound=True, lo=1.0, hi=5.0)
    assert p.shape == (50,) and np.all((p >= 1.0) & (p <= 5.0))
    # `--early-stop MAE|RMSE` for a sharded recommender: the test tuples are routed to their owners ONCE and stay on the devices;
    # the resident evaluation equals the per-call one, before and after a further epoch
    g.set_eval_ratings(test.u, test.j, test.ctx, test.r)
    assert g.eval_resident(1.0, 5.0) == e
    g.train_epoch(util.LR)
    assert g.eval_resident(1.0, 5.0) == g.eval_ratings(test.u, test.j, test.ctx, test.r, 1.0, 5.0) != e


def test_group_of_one_is_the_plain_instance_and_matches_the_oracle():
    model, k = "CAMF_CI", 64
    train, test, gm, state = _problem(model, k, seed=9)
    g = _group(model, k, train, gm, state, 1)
    inst = capi.Instance(model, k, train.n_users, train.n_items, train.n_conds)
    inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
    inst.set_ratings(train.u, train.j, train.ctx, train.r, train.ctx_ptr, train.ctx_conds)
    inst.set_states(state)
    orc = util.c_oracle(model, train, k, state, gm)
    lg, lr = g.train(6, util.LR, bold_driver=True)
    li, lri = inst.train(6, util.LR, bold_driver=True)
    assert np.array_equal(lg, li) and np.array_equal(lr, lri)
    for name, a in g.get_states(np.float32).items():
        assert np.array_equal(a, inst.get_state(name, np.float32)), name
    lo = []
    rate = util.LR
    for it in range(6):                                   # IterativeRecommender.updateLRate, bold driver
        lo.append(orc.epoch(rate))
        if it >= 1:
            rate = rate * 1.05 if abs(lo[-2]) > abs(lo[-1]) else rate * 0.5
    np.testing.assert_allclose(lg, lo, rtol=2e-5)
    eo = orc.eval_ratings(test.u, test.j, test.ctx, test.r, 1.0, 5.0)
    eg = g.eval_ratings(test.u, test.j, test.ctx, test.r, 1.0, 5.0)
    assert abs(eo["RMSE"] - eg["RMSE"]) <= 1e-5 and abs(eo["MAE"] - eg["MAE"]) <= 1e-5


def test_group_fp64_shards_equal_merged_oracles():
    """W = 2 in fp64: every shard is an order-exact pass over its users, so the merged model equals two CPU oracles (one per shard)
    merged the same way on the host, to fp64 rounding -- the multi-GPU algorithm checked against the reference's arithmetic."""
    model, k, world = "CAMF_CI", 32, 2
    train, test, gm, state32 = _problem(model, k, seed=13, n_users=300, n_items=90, n=6000)
    state = {n: a.astype(np.float64) for n, a in state32.items()}
    g = _group(model, k, train, gm, state, world, flags=capi.FLAG_STATE_F64)
    oracles, cuts = [], []
    for rank in range(world):
        shard, (lo, hi) = cdist.shard_by_user(train, rank, world)
        st = {n: (a[lo:hi] if n in ("P", "userBias") else a) for n, a in state.items()}
        oracles.append(util.c_oracle(model, shard, k, st, gm))
        cuts.append((lo, hi))
    for _ in range(3):
        lg = g.train_epoch(util.LR)
        start = {n: oracles[0].state[n].copy() for n in ("Q", "icBias")}
        lo_sum = sum(o.epoch(util.LR) for o in oracles)
        merged = {n: start[n] + sum(o.state[n] - start[n] for o in oracles) / world for n in start}
        for o in oracles:
            for n in merged:
                o.state[n][...] = merged[n]
        assert abs(lg - lo_sum) <= 1e-9 * abs(lo_sum)
    got = g.get_states()
    np.testing.assert_allclose(got["Q"].ravel(), oracles[0].state["Q"].ravel(), rtol=0, atol=1e-11)
    np.testing.assert_allclose(got["icBias"].ravel(), oracles[0].state["icBias"].ravel(), rtol=0, atol=1e-11)
    np.testing.assert_allclose(got["P"].ravel(), np.concatenate([o.state["P"].ravel() for o in oracles]), rtol=0, atol=1e-11)


def test_group_rejects_serial_chain_models_and_bad_calls():
    with pytest.raises(capi.CmiError):
        capi.Group("CAMF_C", 8, 100, 20, 6, 2, devices=[0, 0], flags=capi.FLAG_SCHED_SERIAL)
    g = capi.Group("CAMF_CI", 8, 100, 20, 6, 2, devices=[0, 0])
    with pytest.raises(capi.CmiError, match="set_ratings first"):
        g.set_state("Q", np.zeros((20, 8)))
    with pytest.raises(capi.CmiError):
        capi.Group("CAMF_CI", 8, 100, 20, 6, 2, devices=[0, 99])


@pytest.mark.parametrize("early", [None, "RMSE"])
def test_cpp_host_shards_flag_runs_the_group_and_prints_its_numbers(tmp_path, early):
    """`carskit-mi355x -c setting.conf --shards 2`: the product host trains ONE recommender over a cmi_group (on this box both shards share
    the GPU: the in-process exchange), steered by its unchanged isConverged() -- with `--early-stop RMSE` through the shards' resident test
    tuples -- then evaluates the copied-back model.  Expected numbers: the same folds and C++ init stream through capi.Group behind the
    Python host mirror (local rate x sqrt(2), the hosts' rule)."""
    import re
    import subprocess
    from tests.test_host_layer import EXE, _depaul_conf, expected_from_oracle

    class GroupEngine:
        def __init__(self, model, k, data, tuples, hp, flags=0, device=0):
            u, j, ctx, r = tuples
            self.g = capi.Group(model, k, data.n_users, data.n_items, data.n_conds, 2, devices=[0, 0], flags=flags)
            self.g.set_hparams(hp["regU"], hp["regI"], hp["regB"], hp["regC"], hp["gm"])
            self.g.set_ratings(u, j, ctx, r, data.ctx_ptr, data.ctx_conds)
            self.g.set_lr_scale(np.sqrt(2.0))

        def set_states(self, st):
            self.g.set_states(st)

        def get_states(self):
            return self.g.get_states()

        def epoch(self, lr):
            return self.g.train_epoch(lr)

        def eval_ratings(self, u, j, ctx, r, lo, hi):
            return self.g.eval_ratings(u, j, ctx, r, lo, hi)

        def set_eval_ratings(self, u, j, ctx, r):
            self.g.set_eval_ratings(u, j, ctx, r)
            self.eval_resident_ready = len(r) > 0

        def eval_resident(self, lo, hi):
            return self.g.eval_resident(lo, hi)

    conf = _depaul_conf(tmp_path)
    txt = open(conf).read().replace("recommender=biasedmf", "recommender=camf_ci")
    if early:
        txt = txt.replace("--test-view all", "--test-view all --early-stop " + early)
    open(conf, "w").write(txt)
    flags = capi.FLAG_STATE_F64
    want = expected_from_oracle(conf, "camf_ci", 12, engine_factory=lambda *a, **kw: GroupEngine(*a, **dict(kw, flags=flags)))
    p = subprocess.run([EXE, "-c", conf, "--iters", "12", "--flags", str(flags), "--precise", "--shards", "2"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    m = re.search(r"PRECISE CAMF_CI folds=5 MAE=(\S+) RMSE=(\S+)", p.stdout)
    assert m, p.stdout[-400:] + p.stderr
    assert abs(float(m.group(1)) - want["MAE"]) <= 1e-9 and abs(float(m.group(2)) - want["RMSE"]) <= 1e-9
    # and it is NOT the single-GPU run: the merged two-shard model differs from the sequential one
    q = subprocess.run([EXE, "-c", conf, "--iters", "12", "--flags", str(flags), "--precise"], capture_output=True, text=True)
    m1 = re.search(r"PRECISE CAMF_CI folds=5 MAE=(\S+) RMSE=(\S+)", q.stdout)
    assert m1 and abs(float(m1.group(2)) - float(m.group(2))) > 1e-6

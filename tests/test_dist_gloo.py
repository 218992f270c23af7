"""world_size-2 gloo test of the multi-GPU host logic (carskit_amd/dist.py) on CPU.

The exchange algebra (shard by user -> local order-exact epoch -> sum of item-side deltas -> summed
loss) is exercised with the CPU oracle standing in for the GPU engine (the oracle is the CHECKER's
engine here; the product engine is dist.GpuEngine).  Every rank also simulates all ranks in-process
and demands bit-equality, so a wrong bucket offset, a missed container or a stale epoch-start
snapshot cannot pass."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

from carskit_amd import dist as cdist, synth
from tests import util

MODELS = ["BiasedMF", "PMF", "CAMF_CI", "CAMF_CU", "CAMF_CUCI"]
EPOCHS = 3


class OracleEngine(cdist.TorchEngineMixin):
    """Engine protocol over the CPU oracle (tests only)."""

    def __init__(self, model, shard, k, state, gm):
        self.orc = util.c_oracle(model, shard, k, state, gm)
        self.item = {n: torch.from_numpy(self.orc.state[n]) for n in cdist.ITEM_SIDE[model]}  # shares memory

    def epoch_local(self, lr):
        return self.orc.epoch(lr)

    def item_state(self):
        return self.item


def _shard_state(model, data, k, lo, hi):
    st = synth.init_state(model, data, k, seed=77)      # same global init on every rank
    out = {}
    for n, a in st.items():
        out[n] = a[lo:hi].copy() if n in ("P", "userBias", "ucBias") else a.copy()
    return out


def _shard_for(model, data, rank, world):
    shard, (lo, hi) = cdist.shard_by_user(data, rank, world)   # BiasedMF: util.tuples_for makes the shard's 2-D matrix
    return shard, lo, hi


def _simulate(model, data, k, gm, world, lr, merge="mean"):
    """Single-process restatement of the sharded algorithm (all ranks in turn, deltas summed in rank order)."""
    scale = 1.0 / world if merge == "mean" else 1.0
    engines = []
    for r in range(world):
        shard, lo, hi = _shard_for(model, data, r, world)
        engines.append(OracleEngine(model, shard, k, _shard_state(model, data, k, lo, hi), gm))
    losses = []
    for _ in range(EPOCHS):
        start = {n: engines[0].item[n].clone() for n in engines[0].item}
        ls = [e.epoch_local(lr) for e in engines]
        for n in start:
            assert world == 2
            delta = (engines[0].item[n] - start[n]) + (engines[1].item[n] - start[n])
            merged = start[n] + scale * delta
            for e in engines:
                e.item[n].copy_(merged)
        losses.append(ls[0] + ls[1])
    return engines, losses


def _worker(rank, world, port, model, k, tmpdir, merge="mean"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = util.small_data(n_users=60, n_items=17, n_dims=2, conds_per_dim=3, n=900, seed=31)
        gm = cdist.global_mean(tdist, cdist.shard_by_user(data, rank, world)[0].r)
        assert abs(gm - float(data.r.sum() / np.count_nonzero(data.r))) < 1e-12
        gm = float(data.r.sum() / np.count_nonzero(data.r))
        shard, lo, hi = _shard_for(model, data, rank, world)
        eng = OracleEngine(model, shard, k, _shard_state(model, data, k, lo, hi), gm)
        runner = cdist.ShardedEpochRunner(eng, tdist, merge=merge)
        got_losses = [runner.epoch(util.LR) for _ in range(EPOCHS)]
        sim, want_losses = _simulate(model, data, k, gm, world, util.LR, merge)
        assert got_losses == want_losses, (got_losses, want_losses)
        for n, a in eng.orc.state.items():
            if a is not None:
                assert np.array_equal(a, sim[rank].orc.state[n]), n
        # item side identical on every rank after the exchange
        for n in cdist.ITEM_SIDE[model]:
            t = eng.item[n].clone()
            tdist.broadcast(t, src=0)
            assert torch.equal(t, eng.item[n]), n
        open(os.path.join(tmpdir, "ok%d" % rank), "w").write("ok")
    finally:
        tdist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("model", MODELS)
def test_sharded_epoch_world2_gloo(model, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), model, 6, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_sharded_epoch_world2_gloo_sum_rule(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), "CAMF_CI", 6, str(tmp_path), "sum"), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_shard_by_user_partitions_everything():
    data = util.small_data(n_users=50, n_items=9, n=700, seed=5)
    for world in (1, 2, 3, 8):
        seen = 0
        prev_hi = 0
        sizes = []
        for r in range(world):
            shard, (lo, hi) = cdist.shard_by_user(data, r, world)
            assert lo == prev_hi and hi > lo
            prev_hi = hi
            seen += shard.n
            sizes.append(shard.n)
            assert shard.u.min(initial=0) >= 0 and (shard.n == 0 or shard.u.max() < hi - lo)
            # CRS order kept: positions ascending in the original stream
            idx = np.flatnonzero((data.u >= lo) & (data.u < hi))
            assert np.array_equal(shard.j, data.j[idx]) and np.array_equal(shard.r, data.r[idx])
        assert prev_hi == data.n_users and seen == data.n
        if world in (2, 3):
            assert max(sizes) - min(sizes) <= 0.25 * data.n   # balanced by ratings, not users


def test_world1_runner_is_plain_epoch():
    data = util.small_data(n_users=30, n_items=8, n=300, seed=6)
    gm = float(data.r.sum() / np.count_nonzero(data.r))
    st = synth.init_state("CAMF_CI", data, 5, seed=1)
    a = OracleEngine("CAMF_CI", data, 5, st, gm)
    b = util.c_oracle("CAMF_CI", data, 5, st, gm)
    runner = cdist.ShardedEpochRunner(a, None)
    for _ in range(2):
        assert runner.epoch(util.LR) == b.epoch(util.LR)
    for n, arr in b.state.items():
        if arr is not None:
            assert np.array_equal(arr, a.orc.state[n])


def _preflight_worker(rank, world, port, tmpdir, break_rank, where):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, k = "CAMF_CI", 6
        data = util.small_data(n_users=60, n_items=17, n_dims=2, conds_per_dim=3, n=900, seed=31)
        gm = float(data.r.sum() / np.count_nonzero(data.r))
        shard, lo, hi = _shard_for(model, data, rank, world)

        def make_runner(force_torch):
            eng = OracleEngine(model, shard, k, _shard_state(model, data, k, lo, hi), gm)
            if break_rank == rank and force_torch == (where == "comparison"):   # an exchange that completes but leaves this rank with a different item side
                real = eng.apply

                def bad_apply(scale):
                    real(scale)
                    eng.item["Q"][0, 0] += 1.0
                eng.apply = bad_apply
            return cdist.ShardedEpochRunner(eng, tdist)

        res = cdist.preflight_exchange(make_runner, lambda run: {n: t.numpy() for n, t in run.engine.item.items()}, tdist, lr=util.LR)
        # every rank reports the same verdict, whichever rank the fault was on
        if break_rank < 0:
            assert res["ok"] and res["verified"] and res["note"] == "", res
        elif where == "primary":       # unusable: the caller falls back
            assert not res["ok"] and not res["verified"] and "different item-side states" in res["note"], res
        else:                          # the primary exchange is consistent, the comparison disagrees: keep it, say so
            assert res["ok"] and not res["verified"] and ("MISMATCH" in res["note"] or "another rank" in res["note"]), res
        open(os.path.join(tmpdir, "ok%d" % rank), "w").write("ok")
    finally:
        tdist.destroy_process_group()


@pytest.mark.parametrize("break_rank,where", [(-1, "primary"), (1, "primary"), (1, "comparison")])
def test_exchange_preflight_votes_across_ranks(break_rank, where, tmp_path):
    """carskit_amd.dist.preflight_exchange (what bench.py --gpus N runs before its timed epochs): a sound exchange passes on every rank;
    a primary exchange that leaves ONE rank with a different item-side state fails on EVERY rank, so that all of them take the same
    fallback; a disagreement with the torch-issued comparison run is reported but does not discard a consistent primary exchange."""
    mp.spawn(_preflight_worker, args=(2, _free_port(), str(tmp_path), break_rank, where), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(2))

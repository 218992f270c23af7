"""carskit_amd.dist with the GPU engines and world_size 2: two processes share the test box's single GPU and exchange
through gloo (which all-reduces CUDA tensors).  FM: the sharded sweep must equal a 1-process sweep over all ratings to
rounding.  SGD: both ranks must end with the same item-side state, equal to an in-process simulation of the exchange."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
K = 4


def _fm_worker(rank, world, port, tmpdir):
    import torch
    import torch.distributed as tdist
    from carskit_amd import capi, dist as cdist
    from tests import util
    from tests.test_oracle_fm import REGLF, REGLW, fm_init_model
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = util.small_data(n_users=80, n_items=25, n_dims=2, conds_per_dim=3, n=1500, seed=71)
        w0, w, V = fm_init_model(data.n_users, data.n_items, data.n_conds, K, 5)
        shard, (lo, hi) = cdist.shard_by_user(data, rank, world)
        nu, ni = data.n_users, data.n_items
        sel = np.r_[lo:hi, nu:nu + ni + data.n_conds]

        def fm(d, n_users, ww, VV):
            g = capi.FMInstance(K, n_users, ni, d.n_conds, d.n_dims)
            g.set_hparams(REGLW, REGLF, data.n)            # `size` = all ranks' ratings
            g.set_ratings(d.u, d.j, d.ctx, d.r)
            g.set_model(w0, ww, VV)
            g.init()
            return g
        local, full = fm(shard, hi - lo, w[sel], V[sel]), fm(data, nu, w, V)
        run = cdist.ShardedFMRunner(cdist.GpuFMEngine(local, 0), tdist)
        for _ in range(2):
            run.sweep()
            full.sweep()
        local.synchronize()
        lw0, lw, lV = local.get_model()
        fw0, fw, fV = full.get_model()
        assert abs(lw0 - fw0) < 1e-11
        np.testing.assert_allclose(lw, fw[sel], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(lV, fV[sel], rtol=1e-8, atol=1e-11)
        open(os.path.join(tmpdir, "fm%d" % rank), "w").write("ok")
    finally:
        tdist.destroy_process_group()


def _sgd_worker(rank, world, port, tmpdir):
    import torch
    import torch.distributed as tdist
    from carskit_amd import capi, dist as cdist, synth
    from oracle import oracle_c
    from tests import util
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, k = "CAMF_CI", 64
        data = util.small_data(n_users=400, n_items=120, n_dims=3, conds_per_dim=3, n=8000, seed=72)
        gm = oracle_c.global_mean(data.r)
        full_state = synth.init_state(model, data, k, seed=3)

        def make(r):
            shard, (lo, hi) = cdist.shard_by_user(data, r, world)
            st = {n: (a[lo:hi] if n in ("P", "userBias") else a) for n, a in full_state.items()}
            inst = capi.Instance(model, k, hi - lo, data.n_items, data.n_conds)
            inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
            inst.set_ratings(shard.u, shard.j, shard.ctx, shard.r, data.ctx_ptr, data.ctx_conds)
            inst.set_states(st)
            return inst
        mine = make(rank)
        run = cdist.ShardedEpochRunner(mine, tdist, device_index=0)
        # in-process simulation: both shards trained locally from the same item-side start, the MEAN of their moves applied
        sims = [make(r) for r in range(world)]
        losses = []
        for _ in range(3):
            losses.append(run.epoch(util.LR))
            start = {n: sims[0].get_states()[n].copy() for n in ("Q", "icBias")}
            local_losses = [s.train_epoch(util.LR) for s in sims]
            merged = {n: start[n] + sum((s.get_states()[n] - start[n]) for s in sims) / world for n in start}
            for s in sims:
                s.set_states({n: merged[n] for n in merged})
            # (the simulation merges in fp64 on the host, the runner in fp32 on the device: states agree to an fp32 ulp)
            assert abs(losses[-1] - sum(local_losses)) <= 1e-5 * abs(losses[-1])
        got = mine.get_states()
        for n in ("Q", "icBias"):   # fp32 state: the exchange adds the same numbers in a different order
            np.testing.assert_allclose(got[n], sims[0].get_states()[n], rtol=0, atol=2e-6)
        np.testing.assert_allclose(got["P"], sims[rank].get_states()["P"], rtol=0, atol=2e-6)
        open(os.path.join(tmpdir, "sgd%d" % rank), "w").write("ok")
    finally:
        tdist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("worker,tag", [(_fm_worker, "fm"), (_sgd_worker, "sgd")])
def test_two_ranks_on_one_gpu(worker, tag, tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / (tag + "0")).exists() and (tmp_path / (tag + "1")).exists()

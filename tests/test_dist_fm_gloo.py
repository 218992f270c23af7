"""world_size-2 gloo test of carskit_amd.dist.ShardedFMRunner (FM sweep over user-sharded ratings) with a NumPy
engine, plus a check that the NumPy engine itself reproduces the dense order-exact FM oracle."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as tdist
import torch.multiprocessing as mp

from carskit_amd import dist as cdist
from oracle import oracle_c
from tests import util
from tests.fm_np_engine import NumpyFMEngine
from tests.test_oracle_fm import REGLF, REGLW, fm_init_model

K = 3


def _problem():
    data = util.small_data(n_users=30, n_items=9, n_dims=2, conds_per_dim=3, n=400, seed=61)
    return data, fm_init_model(data.n_users, data.n_items, data.n_conds, K, 8)


def test_numpy_engine_matches_dense_oracle():
    data, (w0, w, V) = _problem()
    orc = oracle_c.FMOracle(K, data.n_users, data.n_items, data.n_conds, data.n_dims, data.u, data.j, data.ctx, data.r,
                            w0, w, V, REGLW, REGLF)
    orc.init()
    eng = NumpyFMEngine(K, data.n_users, data.n_items, data.n_conds, data.n_dims, data.u, data.j, data.ctx, data.r,
                        w0, w, V, REGLW, REGLF, data.n)
    np.testing.assert_allclose(eng.err, orc.errors, rtol=0, atol=1e-13)
    run = cdist.ShardedFMRunner(eng, None)
    for _ in range(2):
        orc.sweep()
        run.sweep()
        assert abs(eng.w0 - orc.w0) < 1e-11
        np.testing.assert_allclose(eng.w, orc.w, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(eng.V, orc.V, rtol=1e-8, atol=1e-12)


def _worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data, (w0, w, V) = _problem()
        shard, (lo, hi) = cdist.shard_by_user(data, rank, world)
        nu, ni = data.n_users, data.n_items
        # local model: this rank's users' rows + the replicated item / context rows
        sel = np.r_[lo:hi, nu:nu + ni + data.n_conds]
        eng = NumpyFMEngine(K, hi - lo, ni, data.n_conds, data.n_dims, shard.u, shard.j, shard.ctx, shard.r, w0, w[sel],
                            V[sel], REGLW, REGLF, data.n)
        run = cdist.ShardedFMRunner(eng, tdist)
        ref = NumpyFMEngine(K, nu, ni, data.n_conds, data.n_dims, data.u, data.j, data.ctx, data.r, w0, w, V, REGLW,
                            REGLF, data.n)
        ref_run = cdist.ShardedFMRunner(ref, None)
        for _ in range(2):
            run.sweep()
            ref_run.sweep()
        assert abs(eng.w0 - ref.w0) < 1e-12
        np.testing.assert_allclose(eng.w, ref.w[sel], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(eng.V, ref.V[sel], rtol=1e-10, atol=1e-13)
        open(os.path.join(tmpdir, "ok%d" % rank), "w").write("ok")
    finally:
        tdist.destroy_process_group()


def test_sharded_fm_world2_gloo(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_phase_field_numbering():
    assert [cdist.fm_phase_field(p) for p in range(10)] == [-1, 0, 1, 2, 0, 1, 2, 0, 1, 2]

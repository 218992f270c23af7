"""SVD++ / CAMF_ICS / CAMF_LCS / CAMF_MCS: the serial GPU kernels against the C oracle on the host, same data.
usage (GPU box): python tests/tools/bench_ext_models.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from carskit_amd import capi, synth
from oracle import oracle_c
from tests import util

data = util.small_data(n_users=2000, n_items=1500, n_dims=3, conds_per_dim=4, n=60000, seed=11)
empty = np.array([dim * 4 + 3 for dim in range(3)], dtype=np.int32)
for model in ("SVD++", "CAMF_ICS", "CAMF_LCS", "CAMF_MCS"):
    for k in (10, 64, 128):
        st = synth.init_state(model, data, k, seed=1)
        gm = float(data.r.sum() / np.count_nonzero(data.r))
        if model == "SVD++":      # the 2-D train matrix (users x items, mean over contexts), row-major: what SVDPlusPlus.java iterates
            u, j, r = synth.to_2d(data)
            ctx = None
        else:
            u, j, ctx, r = util.tuples_for(model, data)
        inst = capi.Instance(model, k, data.n_users, data.n_items, data.n_conds, flags=capi.FLAG_SCHED_SERIAL)
        inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
        if model == "SVD++":
            inst.set_ratings(u, j, None, r)
        else:
            inst.set_sim_params(10, data.n_dims, empty)
            inst.set_ratings(u, j, ctx, r, data.ctx_ptr, data.ctx_conds)
        inst.set_states(st)
        inst.train_epoch(0.01)
        t0 = time.time()
        for _ in range(3):
            inst.train_epoch(0.01)
        g = (time.time() - t0) / 3
        orc = oracle_c.SimOracle(model, k, data.n_users, data.n_items, data.n_conds, u, j, ctx, r, data.ctx_ptr, data.ctx_conds,
                                 empty, {n: np.array(a, np.float64) for n, a in st.items()}, gm,
                                 util.REG, util.REG, util.REG, util.REGC, n_ctx_dims=max(1, data.n_dims))
        t0 = time.time()
        orc.epoch(0.01)
        c = time.time() - t0
        print("%-9s k %3d n %d: GPU %.3f us/tuple (%.2f M updates/s) | C oracle, 1 core %.3f us/tuple (%.2f M/s) | GPU/CPU %.2fx" %
              (model, k, len(r), g / len(r) * 1e6, len(r) / g / 1e6, c / len(r) * 1e6, len(r) / c / 1e6, c / g), flush=True)

"""CAMF_C: the conflict-free-block kernel against the pipelined serial wave (and the one-ahead serial wave) on the same data.
usage (GPU box): python tests/tools/bench_camfc_paths.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from carskit_amd import capi, synth
from tests import util

def run(data, k, env):
    old = {n: os.environ.get(n) for n in ("CMI_NO_CAMFC_BLOCKS", "CMI_NO_CAMFC_PIPE", "CMI_CAMFC_LDS_CHAIN", "CMI_CAMFC_NO_RC")}
    for n in old:
        os.environ.pop(n, None)
    os.environ.update(env)
    try:
        inst = capi.Instance("CAMF_C", k, data.n_users, data.n_items, data.n_conds, flags=capi.FLAG_SCHED_SERIAL)
        inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, float(data.r.mean()))
        inst.set_ratings(data.u, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
        inst.set_states(synth.init_state("CAMF_C", data, k, seed=1))
        inst.train_epoch(0.01)
        t0 = time.time()
        E = 20
        for _ in range(E):
            inst.train_epoch(0.01)
        dt = (time.time() - t0) / E
        info = inst.schedule_info()
        return dt, info.get("flow_blocks", 0)
    finally:
        for n, v in old.items():
            os.environ.pop(n, None)
            if v is not None:
                os.environ[n] = v

for (nu, ni, n, srt) in ((2000, 1500, 60000, False), (2000, 1500, 60000, True), (200, 150, 20000, False), (20000, 8000, 200000, False)):
    data = util.small_data(n_users=nu, n_items=ni, n_dims=3, conds_per_dim=4, n=n, seed=11)
    if srt:
        import dataclasses
        o = np.lexsort((data.j, data.u))
        data = dataclasses.replace(data, u=data.u[o], j=data.j[o], ctx=data.ctx[o], r=data.r[o])
    for k in (10, 64, 128, 256):
        a, nb = run(data, k, {})
        b, _ = run(data, k, {"CMI_NO_CAMFC_BLOCKS": "1"})
        c, _ = run(data, k, {"CMI_NO_CAMFC_BLOCKS": "1", "CMI_NO_CAMFC_PIPE": "1"})
        print("users %d items %d n %d %s k %d: default %.3f us/tuple (blocks %d, %.1f tuples/block) | pipe %.3f | one-ahead %.3f" %
              (nu, ni, data.n, "user-sorted" if srt else "random order", k, a / data.n * 1e6, nb, data.n / nb if nb else 0, b / data.n * 1e6, c / data.n * 1e6), flush=True)

# Frappe shape (BASELINE configs[1]): 957 users x 4 082 items, 8 dimensions / 344 conditions, 96 K ratings -- condBias does not fit a register
data = util.small_data(n_users=957, n_items=4082, n_dims=8, conds_per_dim=43, n=96000, seed=12)
for k in (10, 64, 128, 256):
    a, nb = run(data, k, {})
    b, _ = run(data, k, {"CMI_CAMFC_LDS_CHAIN": "1"})
    c, _ = run(data, k, {"CMI_NO_CAMFC_BLOCKS": "1"})
    print("Frappe shape n %d k %d: lean LDS chain %.3f us/tuple = %.2f M updates/s (blocks %d, %.1f tuples/block) | round-2 LDS chain %.3f | serial wave %.3f" %
          (data.n, k, a / data.n * 1e6, data.n / a / 1e6, nb, data.n / nb if nb else 0, b / data.n * 1e6, c / data.n * 1e6), flush=True)

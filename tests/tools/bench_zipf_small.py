#!/usr/bin/env python3
"""SURVEY 8(d)'s heavy-tail stress run: CAMF_CI k=128, 100 K users x 10 K items, 5 M ratings, Zipf(1.1) items (the hottest item holds
about 15 % of the ratings).  GPU epoch (default schedule and, for comparison, the level walk) against the CPU oracle on the same tuples.
usage: tests/tools/bench_zipf_small.py [zipf [users items ratings]]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from carskit_amd import capi, synth  # noqa: E402
from oracle import oracle_c  # noqa: E402
from tests import util  # noqa: E402


def main():
    z = float(sys.argv[1]) if len(sys.argv) > 1 else 1.1
    nu, ni, n = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (100_000, 10_000, 5_000_000)
    data = synth.generate(nu, ni, 4, 8, n, seed=7, item_zipf=z)
    k = 128
    state = synth.init_state("CAMF_CI", data, k, dtype=np.float32)
    gm = oracle_c.global_mean(data.r)
    out = {"workload": "CAMF_CI k=128, %d users x %d items, %d ratings, Zipf(%g) items, hottest item %d ratings"
           % (data.n_users, data.n_items, data.n, z, int(np.bincount(data.j).max()))}
    for name, flags in (("default", 0), ("level_walk", capi.FLAG_NO_OWNER), ("owner_forced", capi.FLAG_SCHED_OWNER)):
        inst = capi.Instance("CAMF_CI", k, data.n_users, data.n_items, data.n_conds, flags=flags)
        inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
        inst.set_ratings(data.u, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
        inst.set_states(state)
        inst.train_epoch(util.LR)
        t0 = time.perf_counter()
        for _ in range(3):
            inst.train_epoch(util.LR)
        dt = (time.perf_counter() - t0) / 3
        info = inst.schedule_info()
        out[name] = {"schedule": info["kind"], "teams": info.get("teams", 0), "ms_per_epoch": dt * 1e3, "updates_per_s": data.n / dt}
    orc = util.c_oracle("CAMF_CI", data, k, {n: np.asarray(a, dtype=np.float64) for n, a in state.items()}, gm)
    t0 = time.perf_counter()
    orc.epoch(util.LR)
    dt = time.perf_counter() - t0
    out["cpu_oracle_1_core"] = {"ms_per_epoch": dt * 1e3, "updates_per_s": data.n / dt}
    out["gpu_over_cpu"] = out["default"]["updates_per_s"] / out["cpu_oracle_1_core"]["updates_per_s"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Experiment: per-level cost of the narrow-run launch vs table size (is it the HBM latency of the cold partner rows?).
Same item skew and tuples per level, user table 10x smaller -> rows come from the Infinity Cache instead of HBM."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carskit_amd import capi, synth  # noqa: E402


def run(n_users, n_items, n, zipf):
    data = synth.generate(n_users, n_items, 4, 8, n, seed=3, item_zipf=zipf)
    _, off = capi.level_schedule(data.u, data.j, data.n_users, data.n_items)
    state = synth.init_state("CAMF_CI", data, 128, seed=1, dtype=np.float32)
    inst = capi.Instance("CAMF_CI", 128, data.n_users, data.n_items, data.n_conds)
    inst.set_hparams(1e-4, 1e-4, 1e-4, 1e-3, float(data.r.mean()))
    inst.set_ratings(data.u, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
    inst.set_states(state)
    inst.train_epoch(0.02)
    t0 = time.perf_counter()
    for _ in range(2):
        inst.train_epoch(0.02)
    dt = (time.perf_counter() - t0) / 2
    nl = len(off) - 1
    print("users %d items %d ratings %d zipf %.1f: %d levels (mean %.1f tuples), %d launches, %.0f ms/epoch = %.2f us per level, %.1f M updates/s"
          % (n_users, n_items, data.n, zipf, nl, data.n / nl, inst.schedule_info()["levels"], dt * 1e3, dt * 1e6 / nl, data.n / dt / 1e6), flush=True)


if __name__ == "__main__":
    run(100_000, 10_000, 20_000_000, 0.8)     # P 51 MB, Q 5 MB: cache-resident
    run(1_000_000, 10_000, 20_000_000, 0.8)   # P 512 MB: partner rows from HBM

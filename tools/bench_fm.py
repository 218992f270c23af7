#!/usr/bin/env python3
"""FM sweep time on one GPU's share of BASELINE config C4 (FM k=64, 5 M users x 500 K items x 64 conditions,
200 M ratings over 8 GPUs -> 625 K users / 25 M ratings per GPU).  usage: tools/bench_fm.py [ratings] [sweeps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carskit_amd import capi, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
    sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    k = 64
    data = synth.generate_fast(625_000, 500_000, 4, 16, n)
    p = data.n_users + data.n_items + data.n_conds
    rng = np.random.default_rng(1)
    g = capi.FMInstance(k, data.n_users, data.n_items, data.n_conds, data.n_dims)
    g.set_hparams(synth.java_float(0.01), synth.java_float(0.02))
    t0 = time.perf_counter()
    g.set_ratings(data.u, data.j, data.ctx, data.r)
    g.set_model(0.0, rng.random(p), 0.1 * rng.standard_normal((p, k)))
    g.init()
    print("setup %.1fs, %d ratings, p=%d" % (time.perf_counter() - t0, data.n, p), flush=True)
    g.sweep()
    t0 = time.perf_counter()
    for _ in range(sweeps):
        g.sweep()
    dt = (time.perf_counter() - t0) / sweeps
    phases = 4 + 3 * k
    lay = g.layout()
    red_u, red_i = g.time_reduce(4 + 3 * (k // 2)) * 1e-3, g.time_reduce(4 + 3 * (k // 2) + 1) * 1e-3
    print("sweep %.1f ms (%d phases, %.1f us/phase): %.1f M rating-sweeps/s; reduce launch %.1f / %.1f us (user / item field) at %.1f B fetched per "
          "rating-phase by the layout's own count (%d / %d slices); whole sweep %.0f GB/s of implementation bytes = %.1f%% of 8 TB/s"
          % (dt * 1e3, phases, dt * 1e6 / phases, data.n / dt / 1e6, red_u * 1e6, red_i * 1e6,
             0.5 * (lay["bytes_reduce_user"] + lay["bytes_reduce_item"]) / data.n, lay["slices_user_order"], lay["slices_item_order"],
             lay["bytes_per_factor"] * (k + 1) / dt / 1e9, 100 * lay["bytes_per_factor"] * (k + 1) / dt / 8e12))


if __name__ == "__main__":
    main()

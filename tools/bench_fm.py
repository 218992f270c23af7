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
    # algorithmic bytes per rating per sweep (SURVEY 8d, fp64 here): (3+3k) passes over errors (8 B r + 8 B w) and, for the
    # 3k factor passes, one Q column entry (8 B r + 8 B w)
    bytes_per_rating = 16 * (3 + 3 * k) + 16 * 3 * k   # the REFERENCE algorithm's compulsory traffic (errors[] + one Q column entry per phase)
    print("sweep %.1f ms (%d phases, %.1f us/phase): %.1f M rating-sweeps/s, %.0f GB/s algorithmic (fp64) = %.1f%% of 8 TB/s"
          % (dt * 1e3, phases, dt * 1e6 / phases, data.n / dt / 1e6, data.n * bytes_per_rating / dt / 1e9,
             100 * data.n * bytes_per_rating / dt / 8e12))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Aggregate throughput of F independent recommender instances (the reference's `cv -p on`: one Java thread per
fold) sharing ONE GPU, each on its own HIP stream/graph: the level launches of different folds interleave and fill
each other's ramp-up/drain gaps.  usage: tools/bench_folds.py [F] [steps]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carskit_amd import capi, synth  # noqa: E402


def main():
    folds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    model, k = "CAMF_CI", 128
    data = synth.generate_fast(1_000_000, 100_000, 4, 8, 50_000_000)
    regs = (synth.java_float(1e-4),) * 3 + (synth.java_float(1e-3),)
    lr, gm = synth.java_float(0.02), float(data.r.mean())
    state = synth.init_state(model, data, k, dtype=np.float32)
    insts = []
    for f in range(folds):
        # every fold trains on ~80% of the tuples (a different 20% held out), like 5-fold CV
        mask = (np.arange(data.n) % 5) != (f % 5)
        tr = data.subset(np.flatnonzero(mask))
        inst = capi.Instance(model, k, data.n_users, data.n_items, data.n_conds)
        inst.set_hparams(*regs, gm)
        inst.set_ratings(tr.u, tr.j, tr.ctx, tr.r, tr.ctx_ptr, tr.ctx_conds)
        inst.set_states(state)
        inst.train_epoch(lr)  # warm-up + graph capture
        insts.append((inst, tr.n))

    def work(inst):
        for _ in range(steps):
            inst.train_epoch(lr)

    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(i,)) for i, _ in insts]
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    total = sum(n for _, n in insts) * steps
    print("folds=%d steps=%d: %.3f G updates/s aggregate, %.2f ms per fold-epoch wall, %.1f%% of 8 TB/s"
          % (folds, steps, total / dt / 1e9, dt / steps * 1e3, 100 * total * 2120 / dt / 8e12))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Hand-off latency of the owner epoch: ONE user rates N distinct items (every tuple of that user's chain belongs to a different owner
when items are owned), so the epoch is N record hand-offs in a row: writer's write-through store -> reader's poll -> its update.
usage: tools/exp_owner_handoff.py [n]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carskit_amd import capi, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    rng = np.random.default_rng(1)
    u = np.zeros(n, dtype=np.int32)
    j = rng.permutation(n).astype(np.int32)          # n distinct items, one rating each, all by user 0
    r = rng.integers(1, 6, n).astype(np.float64)
    out = {}
    for hub in ("item", "user"):
        os.environ["CMI_OWNER_HUB"] = hub
        inst = capi.Instance("BiasedMF", 128, 1, n, 0, flags=capi.FLAG_SCHED_OWNER)
        inst.set_hparams(1e-4, 1e-4, 1e-4, 1e-3, 3.0)
        inst.set_ratings(u, j, None, r)
        state = {"P": rng.standard_normal((1, 128)).astype(np.float32) * 0.1, "Q": rng.standard_normal((n, 128)).astype(np.float32) * 0.1,
                 "userBias": np.zeros(1, np.float32), "itemBias": np.zeros(n, np.float32)}
        inst.set_states(state)
        inst.train_epoch(0.01)
        t0 = time.perf_counter()
        for _ in range(3):
            inst.train_epoch(0.01)
        dt = (time.perf_counter() - t0) / 3
        out[hub + "s_owned"] = {"schedule": inst.schedule_info()["kind"], "ms_per_epoch": dt * 1e3, "us_per_tuple": dt * 1e6 / n}
    print(json.dumps({"tuples_of_the_one_user": n, **out}))


if __name__ == "__main__":
    main()

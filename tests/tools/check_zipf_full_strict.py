#!/usr/bin/env python3
"""One-off parity check at the full size of the heavy-tail bench: CAMF_CI k=128, 1 M users x 100 K items, 50 M ratings, Zipf(0.8) items
(hottest item 1.1 M ratings).  Owner epoch in strict fp64 on the GPU vs the sequential CPU oracle on the same tuples, one epoch:
every container must be BIT-IDENTICAL.  (tests/test_gpu_fullsize.py holds the 10 M-rating version of this; this one takes minutes.)
usage: tests/tools/check_zipf_full_strict.py [zipf]"""
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
    z = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
    t0 = time.perf_counter()
    data = synth.generate(1_000_000, 100_000, 4, 8, 50_000_000, seed=synth.DEFAULT_SEED, item_zipf=z)
    k = 128
    state = synth.init_state("CAMF_CI", data, k)
    gm = oracle_c.global_mean(data.r)
    t_gen = time.perf_counter() - t0
    inst = capi.Instance("CAMF_CI", k, data.n_users, data.n_items, data.n_conds, flags=capi.FLAG_STATE_F64 | capi.FLAG_STRICT)
    inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
    inst.set_ratings(data.u, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
    inst.set_states(state)
    info = inst.schedule_info()
    t0 = time.perf_counter()
    lg = inst.train_epoch(util.LR)
    t_gpu = time.perf_counter() - t0
    orc = util.c_oracle("CAMF_CI", data, k, state, gm)
    t0 = time.perf_counter()
    lo = orc.epoch(util.LR)
    t_cpu = time.perf_counter() - t0
    got = inst.get_states()
    same = {name: bool(np.array_equal(orc.state[name].reshape(a.shape), a)) for name, a in got.items()}
    print(json.dumps({"workload": "CAMF_CI k=128, %d users x %d items, %d ratings, Zipf(%g) items, hottest item %d ratings"
                                  % (data.n_users, data.n_items, data.n, z, int(np.bincount(data.j).max())),
                      "schedule": info["kind"], "owners": info["flow_blocks"], "gpu_strict_fp64_epoch_s": t_gpu, "cpu_oracle_epoch_s": t_cpu,
                      "loss_gpu": lg, "loss_oracle": lo, "loss_rel_diff": abs(lg - lo) / abs(lo), "containers_bit_identical": same,
                      "all_bit_identical": all(same.values()), "generate_s": t_gen}))
    return 0 if all(same.values()) else 1


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Serial-schedule throughput (CAMF_C k=64 on the Frappe-shaped set = BASELINE config C2) vs the CPU oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from carskit_amd import capi, synth  # noqa: E402
from tests import util  # noqa: E402
from tests.test_gpu_realdata import frappe_shaped  # noqa: E402


def main():
    data = frappe_shaped()
    k = 64
    state = synth.init_state("CAMF_C", data, k)
    gm = float(data.r.mean())
    for flags, name in ((capi.FLAG_SCHED_SERIAL, "fp32 serial"), (capi.FLAG_SCHED_SERIAL | capi.FLAG_STATE_F64, "fp64 serial"),
                        (capi.FLAG_SCHED_SERIAL | capi.FLAG_STATE_F64 | capi.FLAG_STRICT, "fp64 strict serial")):
        inst = capi.Instance("CAMF_C", k, data.n_users, data.n_items, data.n_conds, flags=flags)
        inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
        inst.set_ratings(data.u, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
        inst.set_states(state)
        inst.train_epoch(util.LR)
        name += " (%d blocks)" % inst.schedule_info()["flow_blocks"] if inst.schedule_info()["flow_blocks"] else ""
        t0 = time.perf_counter()
        for _ in range(3):
            inst.train_epoch(util.LR)
        dt = (time.perf_counter() - t0) / 3
        print("%-34s %.1f ms/epoch  %.2f M updates/s" % (name, dt * 1e3, data.n / dt / 1e6), flush=True)
    orc = util.c_oracle("CAMF_C", data, k, state, gm)
    t0 = time.perf_counter()
    for _ in range(3):
        orc.epoch(util.LR)
    dt = (time.perf_counter() - t0) / 3
    print("%-20s %.1f ms/epoch  %.2f M updates/s (1 CPU core)" % ("oracle", dt * 1e3, data.n / dt / 1e6))


if __name__ == "__main__":
    main()

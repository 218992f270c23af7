#!/usr/bin/env python3
"""Latency model of the owner-dataflow epoch on a synthetic workload (tools/micro/owner_sim.c): what a persistent kernel whose
16-lane groups own the heavy-tailed side's rows could reach, before writing it.
usage: tools/owner_sim.py <users> <items> <ratings> <item_zipf> [groups]"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carskit_amd import synth  # noqa: E402


def main():
    nu, ni, n, z = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    groups = int(sys.argv[5]) if len(sys.argv) > 5 else 8192
    so = "/tmp/owner_sim.so"
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools/micro/owner_sim.c")])
    lib = ctypes.CDLL(so)
    data = synth.generate(nu, ni, 4, 8, n, seed=7, item_zipf=z or None)
    hub = np.ascontiguousarray(data.j, dtype=np.int32)
    spoke = np.ascontiguousarray(data.u, dtype=np.int32)
    out = (ctypes.c_double * 4)()
    ip = ctypes.POINTER(ctypes.c_int32)
    print("n=%d users=%d items=%d max item degree=%d" % (data.n, data.n_users, data.n_items, np.bincount(hub).max()))
    for D, L_load, L_pub, t_same, t_switch in [(4, 1.5, 3.5, 0.15, 0.3), (8, 1.5, 3.5, 0.15, 0.3), (16, 1.5, 3.5, 0.15, 0.3),
                                               (8, 2.0, 5.0, 0.2, 0.4), (16, 2.0, 5.0, 0.2, 0.4)]:
        lib.owner_sim(ctypes.c_int64(data.n), hub.ctypes.data_as(ip), spoke.ctypes.data_as(ip), data.n_items, data.n_users, groups, D,
                      ctypes.c_double(L_load), ctypes.c_double(L_pub), ctypes.c_double(t_same), ctypes.c_double(t_switch), out)
        print("D=%2d L_load=%.1f L_pub=%.1f t=%.2f/%.2f us: epoch %.1f ms = %.1f M updates/s (hottest owner: %d tuples, %.1f ms waiting)"
              % (D, L_load, L_pub, t_same, t_switch, out[0] / 1e3, data.n / out[0], out[1], out[3] / 1e3))


if __name__ == "__main__":
    main()

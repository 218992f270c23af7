#!/usr/bin/env python3
"""Differential fuzz of the SVD++ team kernel (svdpp_team.hip) against the C restatement on random 2-D train matrices: random k (incl.
rows longer than the team's factor threads allow for the small team sizes), users with 1 ... hundreds of items (beyond the LDS budget at
large k: the in-kernel fallback), fp32 / fp64 state, 1-3 epochs.   usage: tests/tools/fuzz_svdpp.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from carskit_amd import capi  # noqa: E402
from oracle import oracle_c  # noqa: E402
from tests import util  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = fallbacks = 0
    for case in range(n_cases):
        k = int(rng.choice([1, 3, 10, 16, 17, 40, 64, 100, 128, 200, 256, 300]))
        nu, ni = int(rng.integers(1, 60)), int(rng.integers(2, 400))
        heavy = rng.random() < 0.3
        u, j = [], []
        for x in range(nu):
            m = int(rng.integers(1, min(ni, 400 if heavy and rng.random() < 0.2 else 40) + 1))
            items = np.sort(rng.choice(ni, size=m, replace=False))
            u += [x] * m
            j += items.tolist()
            if m * k * 4 > 140 * 1024:
                fallbacks += 1
        u, j = np.array(u, np.int32), np.array(j, np.int32)
        r = rng.integers(1, 6, size=len(u)).astype(np.float64)
        f64 = bool(rng.integers(2))
        st = {"P": 0.1 * rng.standard_normal((nu, k)), "Q": 0.1 * rng.standard_normal((ni, k)), "userBias": 0.1 * rng.standard_normal(nu),
              "itemBias": 0.1 * rng.standard_normal(ni), "Y": 0.1 * rng.standard_normal((ni, k))}
        gm = float(r.mean())
        z = np.zeros(1, np.int32)
        orc = oracle_c.SimOracle("SVD++", k, nu, ni, 1, u, j, None, r, z, np.zeros(0, np.int32), np.zeros(0, np.int32),
                                 {n: a.copy() for n, a in st.items()}, gm, util.REG, util.REG, util.REG, util.REGC, n_ctx_dims=1)
        inst = capi.Instance("SVD++", k, nu, ni, 1, flags=capi.FLAG_SCHED_SERIAL | (capi.FLAG_STATE_F64 if f64 else 0))
        inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
        inst.set_ratings(u, j, None, r)
        inst.set_states(st)
        ok = True
        lr = util.LR / 8
        for _ in range(int(rng.integers(1, 4))):
            lo, lg = orc.epoch(lr), inst.train_epoch(lr)
            ok &= bool(abs(lo - lg) <= (1e-10 if f64 else 5e-5) * max(1.0, abs(lo)))
        for name, a in inst.get_states().items():
            ok &= bool(np.max(np.abs(orc.state[name].reshape(a.shape) - a), initial=0.0) <= (1e-9 if f64 else 5e-4))
        if not ok:
            bad += 1
            print("MISMATCH case %d: k=%d users=%d items=%d n=%d f64=%s" % (case, k, nu, ni, len(u), f64), flush=True)
    print("%d SVD++ cases (%d users beyond the LDS budget), %d mismatches" % (n_cases, fallbacks, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Differential fuzz of the GPU path against the C oracle on random small problems (every model, random k / dims / sizes /
item skew / flags, incl. the hub-chain kernels forced on both hub sides -- narrow data sends them through sgd_chain_tail -- and the
owner (dataflow) kernel with few / many owners).
fp64+strict: model state must be bit-identical; fp32: loss within 2e-5 and state within 2e-4.
usage: tests/tools/fuzz_gpu.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from carskit_amd import capi, synth  # noqa: E402
from oracle import oracle_c  # noqa: E402
from tests import util  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    F64, SERIAL, STRICT, NOGRAPH = capi.FLAG_STATE_F64, capi.FLAG_SCHED_SERIAL, capi.FLAG_STRICT, capi.FLAG_NO_GRAPH
    CHAIN, OWNER = capi.FLAG_SCHED_CHAIN, capi.FLAG_SCHED_OWNER
    bad = chained = owned = 0
    for case in range(n_cases):
        model = util.MODELS[rng.integers(len(util.MODELS))]
        k = int(rng.choice([1, 2, 5, 10, 16, 31, 64, 70, 100, 128, 130, 256]))
        n_dims = int(rng.integers(0 if model in util.TWO_D else 1, 7))
        n_users, n_items = int(rng.integers(3, 400)), int(rng.integers(2, 300))
        n = int(rng.integers(1, 6000))
        zipf = float(rng.choice([0, 0, 1.1, 1.5]))
        data = synth.generate(n_users, n_items, n_dims, int(rng.integers(1, 5)), n, seed=int(rng.integers(1 << 30)),
                              item_zipf=zipf or None)
        mode = rng.integers(3)
        flags = (F64 | STRICT) if mode == 0 else (F64 if mode == 1 else 0)
        if model == "CAMF_C" or rng.random() < 0.15:
            flags |= SERIAL
        if rng.random() < 0.2:
            flags |= NOGRAPH
        # the hub-chain kernels wherever one exists for this (model, k, precision): fp32 k < 64 (any) and 64..256 (k % 4 == 0), fp64 k = 32..256 (even)
        chain_ok = (model != "CAMF_C" and not flags & (SERIAL | STRICT) and n_dims <= 16 and
                    ((flags & F64 and 32 <= k <= 256 and k % 2 == 0) or (not flags & F64 and (k < 64 or (k <= 256 and k % 4 == 0)))))
        if chain_ok and rng.random() < 0.7:
            flags |= CHAIN
            os.environ["CMI_CHAIN_HUB"] = str(rng.choice(["item", "user", "auto"]))
            os.environ["CMI_CHAIN_MAX"] = str(int(rng.choice([1, 2, 5, 16])))
            chained += 1
        # the owner (dataflow) epoch: every level model, k <= 256 (fp64: 128), <= 384 conditions; few or many owners, either hub side
        owner_ok = (model != "CAMF_C" and not flags & (SERIAL | CHAIN) and k <= (128 if flags & F64 else 256) and
                    (model in util.TWO_D or data.n_conds <= 384))
        os.environ.pop("CMI_OWNER_WAVES", None)
        if owner_ok and rng.random() < 0.4:
            flags |= OWNER
            os.environ["CMI_OWNER_HUB"] = str(rng.choice(["item", "user", "auto"]))
            if rng.random() < 0.5:
                os.environ["CMI_OWNER_WAVES"] = str(int(rng.choice([1, 3, 17, 200])))
            os.environ["CMI_OWNER_TEAM"] = str(rng.choice(["all", "0"]))      # every owner a team of three wavefronts / none
            owned += 1
        state = synth.init_state(model, data, k, seed=int(rng.integers(1 << 30)))
        # CAMF_C (round 3): the pipelined serial wave, the register chain and the lean LDS chain of the block kernel, picked at random
        for knob in ("CMI_NO_CAMFC_BLOCKS", "CMI_NO_CAMFC_PIPE", "CMI_CAMFC_NO_RC", "CMI_CAMFC_LDS_CHAIN"):
            os.environ.pop(knob, None)
        if model == "CAMF_C":
            pick = int(rng.integers(5))
            if pick == 1:
                os.environ["CMI_NO_CAMFC_BLOCKS"] = "1"                                   # the pipelined wave on any order
            elif pick == 2:
                os.environ["CMI_CAMFC_NO_RC"] = "1"                                       # lean LDS chain
            elif pick == 3:
                os.environ["CMI_NO_CAMFC_BLOCKS"], os.environ["CMI_NO_CAMFC_PIPE"] = "1", "1"   # the one-ahead wave
            if rng.random() < 0.3:                                                        # user-sorted input: runs of one
                import dataclasses
                o = np.lexsort((data.j, data.u))
                data = dataclasses.replace(data, u=data.u[o], j=data.j[o], ctx=data.ctx[o], r=data.r[o])
        gm = oracle_c.global_mean(data.r)
        orc = util.c_oracle(model, data, k, state, gm)
        u, j, ctx, r = util.tuples_for(model, data)
        inst = capi.Instance(model, k, data.n_users, data.n_items, data.n_conds, flags=flags)
        inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, gm)
        if model in util.TWO_D:
            inst.set_ratings(u, j, None, r)
        else:
            inst.set_ratings(u, j, ctx, r, data.ctx_ptr, data.ctx_conds)
        inst.set_states(state)
        ok = True
        for _ in range(int(rng.integers(1, 4))):
            lo, lg = orc.epoch(util.LR), inst.train_epoch(util.LR)
            tol = 1e-12 if flags & F64 and flags & STRICT else (1e-9 if flags & F64 else 3e-5)
            if not abs(lo - lg) <= tol * max(1.0, abs(lo)):
                ok = False
        for name, a in inst.get_states().items():
            ref = orc.state[name].reshape(a.shape)
            if flags & F64 and flags & STRICT:
                ok &= bool(np.array_equal(ref, a))
            else:
                ok &= bool(np.max(np.abs(ref - a), initial=0.0) <= (1e-9 if flags & F64 else 3e-4))
        if not ok:
            bad += 1
            print("MISMATCH case %d: %s k=%d dims=%d users=%d items=%d n=%d zipf=%s flags=%#x info=%s"
                  % (case, model, k, n_dims, n_users, n_items, data.n, zipf, flags, inst.schedule_info()), flush=True)
    print("%d cases (%d through the hub-chain kernels, %d through the owner kernel), %d mismatches" % (n_cases, chained, owned, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

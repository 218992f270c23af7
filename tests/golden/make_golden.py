#!/usr/bin/env python3
"""Mint tests/golden/golden_sgd.npz: small seeded inputs and the expected outputs of the CPU oracle
(oracle/carskit_oracle.c, cross-checked by oracle/oracle_np.py) for every SGD model.  The reference itself cannot
run here (Java, no JVM) and ships no expected outputs, so these vectors pin OUR restatement: a change in either
restatement or in the generator shows up as a diff of this file.  Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from carskit_amd import synth  # noqa: E402
from oracle import oracle_c, oracle_np  # noqa: E402
from tests import util  # noqa: E402

K, ITERS = 8, 12


def main():
    data = util.small_data(n_users=40, n_items=15, n_dims=3, conds_per_dim=3, n=500, seed=1234)
    train, test = synth.split(data, 0.2, seed=99)
    out = {"u": train.u, "j": train.j, "ctx": train.ctx, "r": train.r, "ctx_ptr": train.ctx_ptr,
           "ctx_conds": train.ctx_conds, "tu": test.u, "tj": test.j, "tctx": test.ctx, "tr": test.r,
           "dims": np.array([train.n_users, train.n_items, train.n_conds, train.n_dims, K, ITERS]),
           "hparams": np.array([util.LR, util.REG, util.REG, util.REG, util.REGC])}
    gm = oracle_c.global_mean(train.r)
    out["gm"] = np.array([gm])
    for model in util.MODELS:
        state = synth.init_state(model, train, K, seed=4321)
        orc = util.c_oracle(model, train, K, state, gm)
        losses, lrs, _ = orc.build_model(ITERS, util.LR, bold_driver=True)
        # the independent Python restatement must agree before anything is written
        m = util.np_model(model, train, K, state, gm)
        u, j, ctx, r = util.tuples_for(model, train)
        pl, plr = oracle_np.build_model(m, list(zip(u.tolist(), j.tolist(), ctx.tolist(), r.tolist())),
                                        oracle_np.Schedule(util.LR, bold_driver=True), ITERS)
        assert pl == losses.tolist() and plr == lrs.tolist(), model
        tctx = None if model in util.TWO_D else test.ctx
        ev = orc.eval_ratings(test.u, test.j, tctx, test.r, 1.0, 5.0)
        for name, a in state.items():
            out["%s/init/%s" % (model, name)] = a
        for name, a in orc.state.items():
            if a is not None:
                out["%s/final/%s" % (model, name)] = a
        out[model + "/losses"] = losses
        out[model + "/lrates"] = lrs
        out[model + "/eval"] = np.array([ev["MAE"], ev["RMSE"], ev["NMAE"], ev["rMAE"], ev["rRMSE"], ev["n"]])
    path = os.path.join(ROOT, "tests", "golden", "golden_sgd.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

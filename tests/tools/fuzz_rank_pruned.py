#!/usr/bin/env python3
"""Differential fuzz of the tile-pruning walk of the split selection (rank_topn_split_pruned: bounds from the contractions' epilogues,
scalar exclusion cursor, whole-tile loads) against the plain split selection (CMI_RANK_NO_PRUNE=1) on random problems with MANY tiles of
64 candidates and ragged last tiles (the other ranking fuzzers stay below 11 tiles).  The two forms must agree ENTRY FOR ENTRY -- items,
scores, list lengths, measures.  Cases stress what pruning and the unbounded tile loads could get wrong: heavily quantised factors
(tied scores, ties across tiles and with the N-th best), all-equal rows, NaN / +-inf / huge parameters (bounds that are NaN or inf),
thresholds above most scores (short lists), long exclusion lists, lists up to 64, candidate counts that are not multiples of 64, random
batch sizes (a batch's last row borders the slab's slack).
usage (GPU box): tests/tools/fuzz_rank_pruned.py [n_cases] [seed]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from carskit_amd import capi, synth  # noqa: E402


def run(inst, train, test, kw, plain):
    os.environ.pop("CMI_RANK_NO_PRUNE", None)
    if plain:
        os.environ["CMI_RANK_NO_PRUNE"] = "1"
    try:
        return inst.eval_rankings(train, test, **kw)
    finally:
        os.environ.pop("CMI_RANK_NO_PRUNE", None)


def same(a, b):
    return a == b or (isinstance(a, float) and isinstance(b, float) and math.isnan(a) and math.isnan(b))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for case in range(n_cases):
        model = str(rng.choice(["CAMF_CI", "CAMF_CUCI", "CAMF_CU", "BiasedMF", "PMF"]))
        n_users, n_items = int(rng.integers(3, 60)), int(rng.integers(200, 14000))
        d = synth.generate(n_users, n_items, int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(2000, 30000)), seed=int(rng.integers(1 << 30)))
        mask = rng.random(d.n) < float(rng.choice([0.5, 0.8, 0.95]))
        if mask.all() or not mask.any():
            continue
        train = (d.u[mask], d.j[mask], d.ctx[mask], d.r[mask])
        test = (d.u[~mask], d.j[~mask], d.ctx[~mask], d.r[~mask])
        k = int(rng.choice([2, 4, 16, 32]))
        inst = capi.Instance(model, k, d.n_users, d.n_items, d.n_conds)
        inst.set_hparams(1e-4, 1e-4, 1e-4, 1e-3, 0.0 if model == "PMF" else 3.0)
        st = synth.init_state(model, d, k, seed=int(rng.integers(1 << 30)), dtype=np.float32)
        kind = str(rng.choice(["smooth", "quantised", "quantised", "coarse", "flat", "special"]))
        for name, a in st.items():
            if kind == "quantised":                      # many exactly tied scores
                a[...] = np.round(a * 8.0) / 8.0
            elif kind == "coarse":                       # a handful of distinct scores per row
                a[...] = np.sign(a) * 0.5
            elif kind == "flat":                         # every score of a row equal
                a[...] = 0.0
            elif kind == "special" and a.size:
                flat = a.reshape(-1)
                idx = rng.choice(flat.size, size=max(1, flat.size // 500), replace=False)
                flat[idx] = rng.choice(np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 0.0], dtype=np.float32), size=idx.size)
        if model in ("BiasedMF", "PMF"):
            inst.set_ratings(train[0], train[1], None, train[3])
        else:
            inst.set_ratings(train[0], train[1], train[2], train[3], d.ctx_ptr, d.ctx_conds)
        inst.set_states(st)
        kw = dict(bin_thold=float(rng.choice([-1.0, 2.5, 3.2, 3.6])), num_recs=int(rng.choice([1, 3, 10, 25, 64])), num_ignore=int(rng.choice([0, 0, 3])),
                  strategy=str(rng.choice(["ucu", "uc"])), with_lists=True)
        os.environ["CMI_RANK_BATCH"] = str(int(rng.choice([1, 7, 100000])))
        os.environ["CMI_HOST_THREADS"] = str(int(rng.choice([1, 3, 16])))
        msg = None
        with np.errstate(all="ignore"):
            plain = run(inst, train, test, kw, True)
            got = run(inst, train, test, kw, False)
        if set(got[1]) != set(plain[1]):
            msg = "different queries"
        else:
            for key, lst in plain[1].items():
                other = got[1][key]
                if len(other) != len(lst) or any(ia != ib or not same(sa, sb) for (ia, sa), (ib, sb) in zip(lst, other)):
                    msg = "list of %s differs: %r vs %r" % (key, other[:4], lst[:4])
                    break
            if not msg:
                for m, v in plain[0].items():
                    if not same(v, got[0][m]):
                        msg = "measure %s %r vs %r" % (m, got[0][m], v)
                        break
        if msg:
            bad += 1
            print("case %d (%s k=%d %d items %s %s batch %s): %s" % (case, model, k, n_items, kind, kw, os.environ["CMI_RANK_BATCH"], msg), flush=True)
        inst.close()
    print("fuzz_rank_pruned: %d cases, %d bad" % (n_cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Differential fuzz of the SPLIT form of cmi_eval_rankings (fp32 state, MF family) against the per-query form (CMI_RANK_NO_SPLIT=1) on random
small problems with sparse item ids, random batch sizes and host thread counts.  Hard invariants per case: the same queries, the same
list lengths, no list holds an item its user rated in that context in the training set, no list holds an item twice; scores of the two
forms within 1e-5 at equal rank (another association of the same fp32 sum).
usage (GPU box): tests/tools/fuzz_rank_split.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from carskit_amd import capi, synth  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for case in range(n_cases):
        model = str(rng.choice(["CAMF_CI", "CAMF_CUCI", "CAMF_CU", "CAMF_C", "BiasedMF", "PMF"]))
        n_users, n_items = int(rng.integers(3, 150)), int(rng.integers(5, 700))
        d = synth.generate(n_users, n_items, int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(50, 6000)), seed=int(rng.integers(1 << 30)))
        id_space = int(n_items * rng.choice([1, 1, 3, 17]))
        ids = np.sort(rng.choice(id_space, size=d.n_items, replace=False)).astype(np.int32) if id_space > d.n_items else np.arange(d.n_items, dtype=np.int32)
        rng.shuffle(ids)
        j = ids[d.j]
        mask = rng.random(d.n) < 0.75
        if mask.all() or not mask.any():
            continue
        train = (d.u[mask], j[mask], d.ctx[mask], d.r[mask])
        test = (d.u[~mask], j[~mask], d.ctx[~mask], d.r[~mask])
        k = int(rng.choice([4, 16, 33, 64]))
        flags = capi.FLAG_SCHED_SERIAL if model == "CAMF_C" else 0
        inst = capi.Instance(model, k, d.n_users, id_space, d.n_conds, flags=flags)
        inst.set_hparams(1e-4, 1e-4, 1e-4, 1e-3, 0.0 if model == "PMF" else 3.0)
        st = synth.init_state(model, d, k, seed=int(rng.integers(1 << 30)), dtype=np.float32)
        full = {}
        for name, a in st.items():                       # item-side containers live in the sparse id space
            if name in ("Q", "itemBias", "icBias"):
                b = np.zeros((id_space,) + a.shape[1:], dtype=a.dtype)
                b[ids] = a
                a = b
            full[name] = a
        if model in ("BiasedMF", "PMF"):
            inst.set_ratings(train[0], train[1], None, train[3])
        else:
            inst.set_ratings(train[0], train[1], train[2], train[3], d.ctx_ptr, d.ctx_conds)
        inst.set_states(full)
        kw = dict(bin_thold=float(rng.choice([-1.0, 2.5, 3.2])), num_recs=int(rng.choice([1, 3, 10, 40])), num_ignore=int(rng.choice([0, 0, 3])),
                  strategy=str(rng.choice(["ucu", "uc"])), with_lists=True)
        env = {"CMI_RANK_BATCH": str(int(rng.choice([1, 5, 37, 100000]))), "CMI_HOST_THREADS": str(int(rng.choice([1, 3, 16])))}
        os.environ.update(env)
        os.environ.pop("CMI_RANK_NO_SPLIT", None)
        split = inst.eval_rankings(train, test, **kw)
        os.environ["CMI_RANK_NO_SPLIT"] = "1"
        whole = inst.eval_rankings(train, test, **kw)
        os.environ.pop("CMI_RANK_NO_SPLIT", None)
        rated = {}
        for u, jj, c in zip(*(a.tolist() for a in train[:3])):
            rated.setdefault((u, c), set()).add(jj)
        msg = None
        if set(split[1]) != set(whole[1]) or split[0]["n_queries"] != whole[0]["n_queries"]:
            msg = "different queries"
        else:
            for key, lst in whole[1].items():
                other = split[1][key]
                if len(other) != len(lst):
                    msg = "list length %s" % (key,)
                    break
                for got in (lst, other):
                    its = [i for i, _ in got]
                    if len(set(its)) != len(its) or (set(its) & rated.get(key, set())):
                        msg = "rated or repeated item in the list of %s" % (key,)
                        break
                if msg:
                    break
                for (ia, sa), (ib, sb) in zip(lst, other):
                    if abs(sa - sb) > 1e-5 * max(1.0, abs(sa)):
                        msg = "scores %s: %r vs %r" % (key, sa, sb)
                        break
                if msg:
                    break
        if msg:
            bad += 1
            print("case %d (%s k=%d ids x%d %s %s): %s" % (case, model, k, id_space // max(1, d.n_items), kw, env, msg), flush=True)
        inst.close()
    print("fuzz_rank_split: %d cases, %d bad" % (n_cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

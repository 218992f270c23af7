#!/usr/bin/env python3
"""Mint tests/golden/reference_rank.json by EXECUTING the reference's `Recommender.evalRankings` (build container only).

    python oracle/mint_reference_rank.py [/root/reference]

SURVEY 8(f) N4.  After a few interpreted `buildModel()` epochs (oracle/mint_reference_src.py), `evalRankings()` (Recommender.java:672-955)
runs from source: `rateDao.getUserCtxList / getItemList / getRatingCountByItem` are DataDAO.java's own methods, `ranking(u, j, c)` goes
through the model's own `predict`, `carskit.eval.Measures` (the *At cut-off wrappers) is interpreted from its source, and what it inherits --
`happy.coding.math.Measures.{PrecAt, RecallAt, AUC, AP, nDCG, RR}` --, `happy.coding.io.Lists.sortList` (+ its comparator) and
`happy.coding.math.Stats.mean` execute from the BYTECODE of lib/happy.coding.utils-1.2.6.jar.  java.util.HashMap / HashSet / guava
HashMultimap are stand-ins that iterate in the JDK 8 HashMap order (users, contexts and candidate items are visited in that order, which
fixes tie-breaks of the stable sort and the summation order of the means); `Math.log` = fdlibm's.

Inputs, the trained state and the 21 measures are written as data (doubles as hex)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mint_reference_src as M  # noqa: E402


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    rng = np.random.default_rng(20260929)
    out = {"source": "Recommender.evalRankings of the reference, interpreted from source (see oracle/mint_reference_rank.py)", "cases": []}
    grid = (("CAMF_CU", 9, 14, 2, 2, 110, 4, 2, 10, -1.0, "ucu", 0), ("CAMF_CI", 9, 14, 2, 2, 110, 4, 2, 10, 3.0, "uc", 0),
            ("BiasedMF", 8, 16, 2, 2, 100, 3, 2, 10, -1.0, "ucu", 2), ("CAMF_C", 10, 25, 1, 3, 160, 3, 1, 12, 2.0, "ucu", 0),
            ("PMF", 8, 16, 2, 2, 100, 3, 2, 5, -1.0, "uc", 0), ("CAMF_CUCI", 9, 14, 2, 2, 110, 3, 1, 10, -1.0, "ucu", 1),
            # the similarity models are top-N recommenders by construction (isRankingPred), SVD++ by configuration
            ("CAMF_ICS", 9, 14, 2, 3, 110, 3, 2, 10, -1.0, "ucu", 0), ("CAMF_MCS", 9, 14, 2, 3, 110, 3, 2, 10, -1.0, "uc", 0),
            ("CAMF_LCS", 8, 12, 2, 3, 100, 3, 2, 10, -1.0, "ucu", 0), ("SVD++", 8, 16, 2, 2, 100, 3, 2, 10, -1.0, "ucu", 0))
    for (model, nu, ni, nd, cpd, n, k, iters, num_recs, thold, strategy, ignore) in grid:
        prob = M.problem(rng, nu, ni, nd, cpd, n)
        held = [c for i, c in enumerate(prob["cells"]) if i % 4 == 3]
        prob["cells"] = [c for i, c in enumerate(prob["cells"]) if i % 4 != 3]
        rank = {"test_cells": held, "bin_thold": thold, "num_recs": num_recs, "num_ignore": ignore, "strategy": strategy}
        rec = M.run_model(ref, model, prob, k, iters, seed=int(rng.integers(1 << 30)), rank=rank)
        out["cases"].append(rec)
        ms = rec["eval_rankings"]["measures"]
        print("%-10s topN=%d thold=%s %s ignore=%d: Pre10 %.4f  MAP10 %.4f  AUC10 %.4f  NDCG10 %.4f  (%d statements)"
              % (model, num_recs, thold, strategy, ignore, float.fromhex(ms["Pre10"]), float.fromhex(ms["MAP10"]), float.fromhex(ms["AUC10"]),
                 float.fromhex(ms["NDCG10"]), rec["eval_rankings"]["statements"]), flush=True)
    # FM.predict as the scorer (FM.java:76-113) -- the reference's evalRankings() is the same method for every recommender
    out["fm_cases"] = []
    for (nu, ni, nd, cpd, n, k, iters, thold) in ((6, 9, 2, 2, 60, 3, 2, -5.0),):
        prob = M.problem(rng, nu, ni, nd, cpd, n)
        held = [c for i, c in enumerate(prob["cells"]) if i % 4 == 3]
        prob["cells"] = [c for i, c in enumerate(prob["cells"]) if i % 4 != 3]
        rec = M.run_fm(ref, prob, k, iters, seed=int(rng.integers(1 << 30)),
                       rank={"test_cells": held, "bin_thold": thold, "num_recs": 10, "num_ignore": 0, "strategy": "ucu"})
        out["fm_cases"].append(rec)
        ms = rec["eval_rankings"]["measures"]
        print("FM         topN=10: Pre10 %.4f  MAP10 %.4f  AUC10 %.4f  NDCG10 %.4f" % tuple(float.fromhex(ms[x]) for x in ("Pre10", "MAP10", "AUC10", "NDCG10")), flush=True)
    path = os.path.join(ROOT, "tests", "golden", "reference_rank.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path)


if __name__ == "__main__":
    main()

"""Host-only timing of cmi_rank_plan (the plan of cmi_eval_rankings) on the bench's rank workload; no GPU needed.
Prints the best wall time of the sizes-only call (one plan build) and a digest of the plan arrays (must not move with a faster plan)."""
import sys, time, os, hashlib, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from carskit_amd import capi, synth
from carskit_amd.capi import _p, _i64

data = synth.generate_fast(50_000, 20_000, 4, 6, 2_000_000, seed=11)
train, test = synth.split(data, 0.2, seed=3)
c32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
tu, tj, tc, tr = c32(train.u), c32(train.j), c32(train.ctx), np.ascontiguousarray(train.r, dtype=np.float64)
su, sj, sc, sr = c32(test.u), c32(test.j), c32(test.ctx), np.ascontiguousarray(test.r, dtype=np.float64)
sizes = (_i64 * 4)()
args = (train.n_users, train.n_items, len(tu), _p(tu), _p(tj), _p(tc), _p(tr), len(su), _p(su), _p(sj), _p(sc), _p(sr), 2.5, int(sys.argv[1]) if len(sys.argv) > 1 else 0, sizes)
L = capi.lib()
best = 1e9
for _ in range(8):
    t0 = time.perf_counter()
    assert L.cmi_rank_plan(*args, None, None, None, None, None, None, None) == 0
    best = min(best, time.perf_counter() - t0)
nc, nq, nt, ne = list(sizes)
cand, qu, qc = np.zeros(max(nc, 1), np.int32), np.zeros(max(nq, 1), np.int32), np.zeros(max(nq, 1), np.int32)
tp, ti = np.zeros(nq + 1, np.int64), np.zeros(max(nt, 1), np.int32)
ep, ei = np.zeros(nq + 1, np.int64), np.zeros(max(ne, 1), np.int32)
assert L.cmi_rank_plan(*args, _p(cand), _p(qu), _p(qc), _p(tp), _p(ti), _p(ep), _p(ei)) == 0
h = hashlib.sha1()
for a in (cand, qu, qc, tp, ti, ep, ei):
    h.update(a.tobytes())
print("cmi_rank_plan best %.2f ms; sizes %s; digest %s" % (best * 1e3, [nc, nq, nt, ne], h.hexdigest()[:16]))

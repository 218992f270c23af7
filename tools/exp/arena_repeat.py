"""Experiment: is the north_star epoch's run-to-run spread (56 .. 64 ms) decided per PROCESS or per ALLOCATION?
One process generates the workload once, then builds / times / destroys the instance several times (each time the arena and the
tables are freshly hipMalloc'ed).  Prints the mean HIP-event ms per epoch of every instance."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from carskit_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "northstar"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
model, k, n_users, n_items, n_dims, cpd, n_ratings = bench.WORKLOADS[name]
data = synth.generate_fast(n_users, n_items, n_dims, cpd, n_ratings, seed=synth.DEFAULT_SEED)
gm = float(data.r.sum() / np.count_nonzero(data.r))
state = synth.init_state(model, data, k, seed=synth.DEFAULT_SEED + 2, dtype=np.float32)
regs = (synth.java_float(1e-4), synth.java_float(1e-4), synth.java_float(1e-4), synth.java_float(1e-3))
lr = synth.java_float(0.02)
out = []
for rep in range(reps):
    inst = bench.make_instance(model, k, data, n_items, state, regs, gm, 0, 0)
    per = []
    for rnd in range(3):
        _, el, ms = bench.timed_epochs(inst, lr, 4, 1 if rnd == 0 else 0)
        per.append(round(ms, 2))
    out.append(per)
    print(rep, per, inst.schedule_info()["kind"], flush=True)
    inst.close()
print(json.dumps(out))

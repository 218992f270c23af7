"""usage (GPU box): CMI_SETUP_TIMES=1 python tools/exp/setup_time.py [c3|northstar|c5]  -- wall time of Instance creation + cmi_set_ratings +
state upload for one of bench.py's workloads, with cmi_set_ratings' own phases on stderr."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import bench
from carskit_amd import capi, synth

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
model, k, nu, ni, nd, cpd, nr = bench.WORKLOADS[name]
t0 = time.perf_counter()
data = synth.generate_fast(nu, ni, nd, cpd, nr)
t1 = time.perf_counter()
state = synth.init_state(model, data, k, dtype=np.float32)
t2 = time.perf_counter()
inst = capi.Instance(model, k, data.n_users, data.n_items, data.n_conds)
inst.set_hparams(1e-4, 1e-4, 1e-4, 1e-3, float(data.r.mean()))
t3 = time.perf_counter()
inst.set_ratings(data.u, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
t4 = time.perf_counter()
inst.set_states(state)
inst.synchronize()
t5 = time.perf_counter()
l = inst.train_epoch(0.02)
t6 = time.perf_counter()
print("%s: generate %.1f s, init_state %.1f s, create %.2f s, set_ratings %.2f s, set_states %.2f s, first epoch %.3f s (%s)" %
      (name, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, inst.schedule_info()["kind"]))

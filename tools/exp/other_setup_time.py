"""usage (GPU box): python tools/exp/other_setup_time.py  -- wall time of cmi_fm_set_ratings (C4 share, 25 M ratings) + model upload + one sweep, and of
cmi_set_ratings for a 2-D model (BiasedMF) on C3's tuples; CMI_SETUP_TIMES=1 prints cmi_set_ratings' phases."""
import sys, time, numpy as np
sys.path.insert(0,'.')
from carskit_amd import capi, synth
data = synth.generate_fast(625_000, 500_000, 4, 16, 25_000_000)
g = capi.FMInstance(64, data.n_users, data.n_items, data.n_conds, data.n_dims)
g.set_hparams(0.01, 0.02)
t=time.perf_counter(); g.set_ratings(data.u, data.j, data.ctx, data.r); g.synchronize(); print("fm set_ratings %.2f s" % (time.perf_counter()-t))
p = data.n_users + data.n_items + data.n_conds
rng = np.random.default_rng(1)
t=time.perf_counter(); g.set_model(0.0, rng.random(p), 0.1 * rng.standard_normal((p, 64))); g.init(); g.synchronize(); print("set_model+init %.2f s" % (time.perf_counter()-t))
t=time.perf_counter(); g.sweep(); print("sweep %.3f s" % (time.perf_counter()-t))
# plain levels path for BiasedMF (2-D) at 50M
d2 = synth.generate_fast(1_000_000, 100_000, 4, 8, 50_000_000)
st = synth.init_state("BiasedMF", d2, 64, dtype=np.float32)
inst = capi.Instance("BiasedMF", 64, d2.n_users, d2.n_items, d2.n_conds)
inst.set_hparams(1e-4,1e-4,1e-4,1e-3, float(d2.r.mean()))
t=time.perf_counter(); inst.set_ratings(d2.u, d2.j, None, d2.r); print("BiasedMF set_ratings %.2f s" % (time.perf_counter()-t), inst.schedule_info()["kind"])

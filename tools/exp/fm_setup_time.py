"""usage (GPU box): CMI_SETUP_TIMES=1 python tools/exp/fm_setup_time.py  -- cmi_fm_set_ratings on the C4 share (625 K users x 500 K items, 25 M
ratings) with its phases on stderr, then set_model + init and one sweep."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from carskit_amd import capi, synth
data = synth.generate_fast(625_000, 500_000, 4, 16, 25_000_000)
g = capi.FMInstance(64, data.n_users, data.n_items, data.n_conds, data.n_dims)
g.set_hparams(0.01, 0.02)
t = time.perf_counter(); g.set_ratings(data.u, data.j, data.ctx, data.r); g.synchronize(); print("fm set_ratings %.2f s" % (time.perf_counter() - t))
p = data.n_users + data.n_items + data.n_conds
rng = np.random.default_rng(1)
t = time.perf_counter(); g.set_model(0.0, rng.random(p), 0.1 * rng.standard_normal((p, 64))); g.init(); g.synchronize(); print("set_model + init %.2f s" % (time.perf_counter() - t))
t = time.perf_counter(); g.sweep(); g.synchronize(); print("sweep %.3f s" % (time.perf_counter() - t))

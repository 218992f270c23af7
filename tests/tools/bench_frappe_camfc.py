"""CAMF_C on the real Frappe file (BASELINE configs[1]): epoch time of the product path.  usage (GPU box): python tests/tools/bench_frappe_camfc.py"""
import sys, time, tempfile, numpy as np
sys.path.insert(0, '.')
from carskit_amd import capi, dao, synth
from tests import frappe, util
tmp = tempfile.mkdtemp()
src = frappe.write_ratings(tmp, "log")
dao.transform(src, tmp + "/train.csv")
d = dao.DataDAO(tmp + "/train.csv").rating_data()
for k in (64, 256):
    inst = capi.Instance("CAMF_C", k, d.n_users, d.n_items, d.n_conds, flags=capi.FLAG_SCHED_SERIAL)
    inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, float(d.r.mean()))
    inst.set_ratings(d.u, d.j, d.ctx, d.r, d.ctx_ptr, d.ctx_conds)
    inst.set_states(synth.init_state("CAMF_C", d, k, seed=1))
    inst.train_epoch(0.01)
    t0 = time.time(); E = 20
    for _ in range(E): inst.train_epoch(0.01)
    dt = (time.time() - t0) / E
    print("real Frappe CAMF_C k=%d: %.3f ms/epoch = %.2f M updates/s (%.3f us per tuple), schedule %s" % (k, dt * 1e3, d.n / dt / 1e6, dt / d.n * 1e6, inst.schedule_info()["kind"]), flush=True)

import time, sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from carskit_amd import capi, dao, synth
d = dao.DataDAO("tests/golden/depaul_ratings_compact.csv") if False else None
from carskit_amd import dao as D
import tempfile
td = tempfile.mkdtemp()
D.transform("tests/golden/depaul_ratings_compact.csv", td + "/train.csv")
d = D.DataDAO(td + "/train.csv")
rd = d.rating_data()
print("depaul", rd.n, rd.n_users, rd.n_items, rd.n_conds)
for model in ("CAMF_CU", "CAMF_CI", "BiasedMF", "CAMF_C"):
    for k in (10, 64):
        flags = capi.FLAG_SCHED_SERIAL if model == "CAMF_C" else 0
        inst = capi.Instance(model, k, rd.n_users, rd.n_items, rd.n_conds, flags=flags)
        gm = float(rd.r.sum() / np.count_nonzero(rd.r))
        inst.set_hparams(0.001, 0.001, 0.001, 0.001, gm)
        if model == "BiasedMF":
            # 2-D: mean over contexts
            inst.set_ratings(rd.u, rd.j, None, rd.r)
        else:
            inst.set_ratings(rd.u, rd.j, rd.ctx, rd.r, rd.ctx_ptr, rd.ctx_conds)
        st = synth.init_state(model, rd, k, seed=1)
        inst.set_states(st)
        inst.train_epoch(0.01)
        t0 = time.time()
        E = 100
        for _ in range(E):
            inst.train_epoch(0.01)
        dt = time.time() - t0
        print(model, k, "epochs/s %.0f  updates/s %.2f M  ms/epoch %.3f" % (E / dt, E * rd.n / dt / 1e6, dt / E * 1e3), inst.schedule_note() if hasattr(inst, "schedule_note") else "")

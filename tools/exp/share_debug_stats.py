"""GPU box: python tools/exp/share_debug_stats.py [k] [team] -- how often is an instance WITH teams, trained beside two other owner
epochs (cmi_set_device_share(3)), not bit-identical to the one-wavefront form?  STATS_REPS (12) runs each in fp32 and fp64.
Round 5: k = 64: fp32 0 / 12, fp64 12 / 12; k = 128: 0 / 12 and 0 / 12 -- cause unknown.  Round 6: a store-data hazard in the record
stores (docs/history/r06.md 1): 23 / 24 on a library built without the wait states, 0 / 24 on the shipped one, every k and dtype."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from carskit_amd import capi, synth
from tests import util
OWNER, F64 = capi.FLAG_SCHED_OWNER, capi.FLAG_STATE_F64
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
TEAM = None if len(sys.argv) < 3 or sys.argv[2] == "default" else sys.argv[2]   # CMI_OWNER_TEAM of the instance under test
# optional (experiment builds): STATS_CUS_TEST / STATS_CUS_NEIGH = CMI_STREAM_CUS of the instance under test / of its neighbours (compute
# unit masks, "m:r0,r1,..."), STATS_NEIGH_WAVES = CMI_OWNER_WAVES of the neighbours, STATS_REPS, STATS_DTYPES = "f32,f64"
REPS = int(os.environ.get("STATS_REPS", "12"))
DTYPES = os.environ.get("STATS_DTYPES", "f32,f64").split(",")
def make(d, team, flags, neighbour=False):
    if team is None: os.environ.pop("CMI_OWNER_TEAM", None)
    else: os.environ["CMI_OWNER_TEAM"] = team
    cus = os.environ.get("STATS_CUS_NEIGH" if neighbour else "STATS_CUS_TEST")
    if cus: os.environ["CMI_STREAM_CUS"] = cus
    else: os.environ.pop("CMI_STREAM_CUS", None)
    waves = os.environ.get("STATS_NEIGH_WAVES") if neighbour else None
    if waves: os.environ["CMI_OWNER_WAVES"] = waves
    else: os.environ.pop("CMI_OWNER_WAVES", None)
    st = synth.init_state("CAMF_CI", d, K, seed=5, dtype=np.float64 if flags & F64 else np.float32)
    i = capi.Instance("CAMF_CI", K, d.n_users, d.n_items, d.n_conds, flags=OWNER | flags)
    i.set_hparams(util.REG, util.REG, util.REG, util.REGC, 3.0)
    i.set_device_share(3)
    i.set_ratings(d.u, d.j, d.ctx, d.r, d.ctx_ptr, d.ctx_conds)
    i.set_states(st)
    return i
ds = [synth.generate(3000, 300, 3, 4, 120000, seed=500 + s, item_zipf=1.2) for s in (1, 2, 3)]
for flags in [f for f, n in ((0, "f32"), (F64, "f64")) if n in DTYPES]:
    ref = make(ds[0], "0", flags)
    for _ in range(3): ref.train_epoch(util.LR)
    a = ref.get_states()
    bad = 0
    for rep in range(REPS):
        conc = [make(ds[0], TEAM, flags), make(ds[1], "0", flags, True), make(ds[2], "0", flags, True)]
        with ThreadPoolExecutor(max_workers=3) as pool:
            list(pool.map(lambda i: [i.train_epoch(util.LR) for _ in range(3)], conc))
        b = conc[0].get_states()
        teams = conc[0].schedule_info().get("teams")
        e = max(float(np.max(np.abs(a[n].astype(np.float64) - b[n].astype(np.float64)))) for n in a)
        bad += e > 0
        for c in conc: c.close()
    print("k", K, "f64" if flags else "f32", "inexact runs", bad, "of", REPS, "(teams in the instance under test: %s)" % teams)

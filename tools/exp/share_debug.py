import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from carskit_amd import capi, synth
from tests import util
from tests.test_gpu_parity import make_pair
OWNER, F64 = capi.FLAG_SCHED_OWNER, capi.FLAG_STATE_F64
def err(orc, inst):
    return max(float(np.max(np.abs(orc.state[n].reshape(a.shape) - a))) for n, a in inst.get_states().items())
for share, flags, team in ((3, F64, None), (3, F64, "0"), (8, F64, None), (16, F64, None), (3, 0, None), (16, 0, None)):
    if team is None: os.environ.pop("CMI_OWNER_TEAM", None)
    else: os.environ["CMI_OWNER_TEAM"] = team
    pairs = []
    for seed in (1, 2, 3):
        d = synth.generate(3000, 300, 3, 4, 120000, seed=500 + seed, item_zipf=1.2)
        pairs.append(make_pair("CAMF_CI", d, 64, flags | OWNER, before_ratings=lambda i: i.set_device_share(share)))
    work = lambda p: [p[1].train_epoch(util.LR) for _ in range(3)]
    with ThreadPoolExecutor(max_workers=3) as pool:
        losses = list(pool.map(work, pairs))
    out = []
    for (orc, inst), ls in zip(pairs, losses):
        lo = [orc.epoch(util.LR) for _ in ls]
        out.append("%.1e" % err(orc, inst))
    print("share", share, "f64" if flags else "f32", "team", team, pairs[0][1].schedule_info()["flow_blocks"], pairs[0][1].schedule_info().get("teams"), out)

"""GPU box:
python tools/exp/share_debug.py -- owner epochs of three instances in flight together (cmi_set_device_share(3)): which forms stay exact?
Prints the largest deviation from the fp64 oracle per instance for combinations of team / one-wavefront instances.  Round 5: the
one-wavefront form is exact in every combination; an instance with teams is ~2e-7 off in most runs when another owner epoch runs beside it
(from its second epoch on, every row), exact alone, after another, and beside level / chain kernels.  Round 6: cause found and fixed
(a store-data hazard in the record stores, docs/history/r06.md 1): every combination is exact."""
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
for teams in (("all", "0", "0"), (None, "0", "0"), (None, None, "0"), ("0", "0", "0"), (None, None, None)):
    pairs = []
    for seed, team in zip((1, 2, 3), teams):
        if team is None: os.environ.pop("CMI_OWNER_TEAM", None)
        else: os.environ["CMI_OWNER_TEAM"] = team
        d = synth.generate(3000, 300, 3, 4, 120000, seed=500 + seed, item_zipf=1.2)
        pairs.append(make_pair("CAMF_CI", d, 64, F64 | OWNER, before_ratings=lambda i: i.set_device_share(3)))
    work = lambda p: [p[1].train_epoch(util.LR) for _ in range(3)]
    with ThreadPoolExecutor(max_workers=3) as pool:
        losses = list(pool.map(work, pairs))
    out = []
    for (orc, inst), ls in zip(pairs, losses):
        lo = [orc.epoch(util.LR) for _ in ls]
        out.append("teams %s err %.1e" % (inst.schedule_info().get("teams"), err(orc, inst)))
    print(teams, out)

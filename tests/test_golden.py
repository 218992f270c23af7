"""Committed golden vectors (tests/golden/golden_sgd.npz, minted by tests/golden/make_golden.py):
  * CPU: the oracle still reproduces them bit for bit (guards the checker itself);
  * GPU: the HIP path reproduces them WITHOUT the oracle at run time -- bit-exact in fp64-strict serial mode,
    within the north_star's 1e-5 in fp32."""
import os

import numpy as np
import pytest

from carskit_amd import capi, synth
from tests import util

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_sgd.npz"))
NU, NI, NC, ND, K, ITERS = [int(x) for x in G["dims"]]
LR, REGU, REGI, REGB, REGC = [float(x) for x in G["hparams"]]
GM = float(G["gm"][0])


def _data():
    return synth.RatingData(NU, NI, NC, ND, G["u"], G["j"], G["ctx"], G["r"], G["ctx_ptr"], G["ctx_conds"])


def _init(model):
    return {k.split("/")[-1]: G[k] for k in G.files if k.startswith(model + "/init/")}


@pytest.mark.parametrize("model", util.MODELS)
def test_oracle_reproduces_golden(model):
    orc = util.c_oracle(model, _data(), K, _init(model), GM, REGU, REGI, REGB, REGC)
    losses, lrs, _ = orc.build_model(ITERS, LR, bold_driver=True)
    assert losses.tolist() == G[model + "/losses"].tolist() and lrs.tolist() == G[model + "/lrates"].tolist()
    for k in G.files:
        if k.startswith(model + "/final/"):
            assert np.array_equal(orc.state[k.split("/")[-1]].reshape(G[k].shape), G[k]), k


def _gpu(model, flags):
    d = _data()
    u, j, ctx, r = util.tuples_for(model, d)
    if model == "CAMF_C":
        flags |= capi.FLAG_SCHED_SERIAL
    inst = capi.Instance(model, K, NU, NI, NC, flags=flags)
    inst.set_hparams(REGU, REGI, REGB, REGC, GM)
    if model in util.TWO_D:
        inst.set_ratings(u, j, None, r)
    else:
        inst.set_ratings(u, j, ctx, r, d.ctx_ptr, d.ctx_conds)
    inst.set_states(_init(model))
    return inst


@pytest.mark.gpu
@pytest.mark.parametrize("model", util.MODELS)
def test_gpu_reproduces_golden(model):
    tctx = None if model in util.TWO_D else G["tctx"]
    want_eval = G[model + "/eval"]
    # fp64, strict order, serial schedule: every number is the golden one, digit for digit
    inst = _gpu(model, capi.FLAG_STATE_F64 | capi.FLAG_STRICT | capi.FLAG_SCHED_SERIAL)
    losses, lrs = inst.train(ITERS, LR, bold_driver=True)
    assert losses.tolist() == G[model + "/losses"].tolist() and lrs.tolist() == G[model + "/lrates"].tolist()
    for name, a in inst.get_states().items():
        assert np.array_equal(a, G["%s/final/%s" % (model, name)].reshape(a.shape)), name
    ev = inst.eval_ratings(G["tu"], G["tj"], tctx, G["tr"], 1.0, 5.0)
    assert abs(ev["RMSE"] - want_eval[1]) <= 1e-12 and abs(ev["MAE"] - want_eval[0]) <= 1e-12 and ev["n"] == want_eval[5]
    # fp32 default schedule: the north_star tolerance
    inst32 = _gpu(model, 0)
    l32, lr32 = inst32.train(ITERS, LR, bold_driver=True)
    assert lr32.tolist() == G[model + "/lrates"].tolist()
    np.testing.assert_allclose(l32, G[model + "/losses"], rtol=2e-5)
    ev32 = inst32.eval_ratings(G["tu"], G["tj"], tctx, G["tr"], 1.0, 5.0)
    assert abs(ev32["RMSE"] - want_eval[1]) <= 1e-5 and abs(ev32["MAE"] - want_eval[0]) <= 1e-5

"""Hand-checkable single-update known answers.  All inputs are dyadic rationals, so every fp64
operation of one SGD update is exact and the expected values can be derived with exact rational
arithmetic straight from the update rules (Baltrunas et al. 2011 CAMF; reference formulas cited in
oracle/carskit_oracle.h) -- a third derivation, independent of evaluation order."""
from fractions import Fraction as F

import numpy as np
import pytest

from oracle import oracle_c
from tests import util

K = 2
GM, R, LR = F(3), F(4), F(1, 2)
REGU, REGI, REGB, REGC = F(1, 4), F(1, 8), F(1, 16), F(1, 2)
P0 = [F(1, 2), F(-1, 4)]
Q0 = [F(1, 4), F(1, 2)]
BU, BJ = F(1, 2), F(-1, 4)
BC = [F(1, 8), F(-1, 2), F(1, 4)]       # condBias for conditions 0..2
BUC = [F(1, 4), F(1, 8), F(-1, 8)]      # ucBias row of user 0
BIC = [F(-1, 4), F(1, 2), F(1, 16)]     # icBias row of item 0
CONDS = [0, 2]                          # the rating's context = conditions {0, 2}


def _expected(model):
    dot = sum(p * q for p, q in zip(P0, Q0))
    pred = (GM + dot) if model != "PMF" else dot
    if model in ("BiasedMF", "CAMF_C", "CAMF_CI"):
        pred += BU
    if model in ("BiasedMF", "CAMF_C", "CAMF_CU"):
        pred += BJ
    for c in CONDS:
        if model == "CAMF_C":
            pred += BC[c]
        if model in ("CAMF_CI", "CAMF_CUCI"):
            pred += BIC[c]
        if model in ("CAMF_CU", "CAMF_CUCI"):
            pred += BUC[c]
    e = R - pred
    loss = e * e
    out = {}
    if model in ("BiasedMF", "CAMF_C", "CAMF_CI"):
        out["userBias"] = [BU + LR * (e - REGB * BU)]
        loss += REGB * BU * BU
    if model in ("BiasedMF", "CAMF_C", "CAMF_CU"):
        out["itemBias"] = [BJ + LR * (e - REGB * BJ)]
        loss += REGB * BJ * BJ
    if model == "CAMF_C":
        out["condBias"] = [b + LR * (e - REGC * b) if c in CONDS else b for c, b in enumerate(BC)]
        loss += REGB * sum(BC[c] for c in CONDS)            # the reference's quirk: plain sum, regB
    if model in ("CAMF_CI", "CAMF_CUCI"):
        out["icBias"] = [b + LR * (e - REGC * b) if c in CONDS else b for c, b in enumerate(BIC)]
        loss += REGC * sum(BIC[c] ** 2 for c in CONDS)
    if model in ("CAMF_CU", "CAMF_CUCI"):
        out["ucBias"] = [b + LR * (e - REGC * b) if c in CONDS else b for c, b in enumerate(BUC)]
        loss += REGC * sum(BUC[c] ** 2 for c in CONDS)
    out["P"] = [p + LR * (e * q - REGU * p) for p, q in zip(P0, Q0)]
    out["Q"] = [q + LR * (e * p - REGI * q) for p, q in zip(P0, Q0)]
    loss += sum(REGU * p * p + REGI * q * q for p, q in zip(P0, Q0))
    return pred, out, loss / 2


def _state(model):
    f = lambda xs: np.array([float(x) for x in xs])
    st = {"P": f(P0).reshape(1, K), "Q": f(Q0).reshape(1, K)}
    if model in ("BiasedMF", "CAMF_C", "CAMF_CI"):
        st["userBias"] = f([BU])
    if model in ("BiasedMF", "CAMF_C", "CAMF_CU"):
        st["itemBias"] = f([BJ])
    if model == "CAMF_C":
        st["condBias"] = f(BC)
    if model in ("CAMF_CI", "CAMF_CUCI"):
        st["icBias"] = f(BIC).reshape(1, 3)
    if model in ("CAMF_CU", "CAMF_CUCI"):
        st["ucBias"] = f(BUC).reshape(1, 3)
    return st


@pytest.mark.parametrize("model", util.MODELS)
def test_single_update(model):
    pred, exp, loss = _expected(model)
    i32 = lambda *v: np.array(v, dtype=np.int32)
    orc = oracle_c.Oracle(model, K, 1, 1, 3, i32(0), i32(0), i32(0), np.array([float(R)]), i32(0, 2), i32(*CONDS),
                          _state(model), float(GM), float(REGU), float(REGI), float(REGB), float(REGC))
    assert orc.predict(0, 0, 0) == float(pred)
    got_loss = orc.epoch(float(LR))
    assert got_loss == float(loss)
    for name, vals in exp.items():
        assert orc.state[name].reshape(-1).tolist() == [float(v) for v in vals], name


def test_camf_ci_numbers_spelled_out():
    """One model written out digit by digit (so a reader can check it with pencil and paper):
    dot = .5*.25 + (-.25)*.5 = 0 ; pred = 3 + .5 + 0 + (-.25) + .0625 = 3.3125 ; e = .6875."""
    pred, exp, loss = _expected("CAMF_CI")
    assert pred == F(53, 16)
    e = F(11, 16)
    assert exp["userBias"] == [F(1, 2) + F(1, 2) * (e - F(1, 32))]          # 0.828125
    assert exp["icBias"] == [F(-1, 4) + F(1, 2) * (e + F(1, 8)), F(1, 2), F(1, 16) + F(1, 2) * (e - F(1, 32))]
    assert exp["P"] == [F(1, 2) + F(1, 2) * (e / 4 - F(1, 8)), F(-1, 4) + F(1, 2) * (e / 2 + F(1, 16))]
    assert float(exp["userBias"][0]) == 0.828125

"""Shared helpers for the test-suite (problem builders, oracle adapters)."""
import numpy as np

from carskit_amd import synth
from oracle import oracle_np

MODELS = ["BiasedMF", "CAMF_C", "CAMF_CI", "CAMF_CU", "CAMF_CUCI", "PMF"]
TWO_D = ("BiasedMF", "PMF")   # recommenders that iterate the 2-D (user x item) train matrix

# setting.conf defaults as the doubles the reference computes with (Java float -> double)
REG = synth.java_float(1e-4)
REGC = synth.java_float(1e-3)
LR = synth.java_float(2e-2)


def small_data(n_users=23, n_items=11, n_dims=2, conds_per_dim=3, n=300, seed=7, item_zipf=None):
    return synth.generate(n_users, n_items, n_dims, conds_per_dim, n, seed=seed, item_zipf=item_zipf)


def tuples_for(model, data):
    """(u, j, ctx, r) arrays in the order the model's buildModel iterates."""
    if model in TWO_D:
        u, j, r = synth.to_2d(data)
        return u, j, np.zeros(len(r), np.int32), r
    return data.u, data.j, data.ctx, data.r


def np_model(model, data, k, state, gm, regU=REG, regI=REG, regB=REG, regC=REGC):
    conds = [data.ctx_conds[data.ctx_ptr[c]:data.ctx_ptr[c + 1]].tolist() for c in range(data.n_ctx)]
    m = oracle_np.MODELS[model](k, data.n_users, data.n_items, data.n_conds, conds, gm, regU, regI, regB, regC)
    for name, a in state.items():
        setattr(m, name, np.asarray(a, dtype=np.float64).tolist())
    return m


def np_state(m):
    out = {}
    for name in ("P", "Q", "userBias", "itemBias", "condBias", "ucBias", "icBias"):
        v = getattr(m, name)
        if v is not None:
            out[name] = np.array(v, dtype=np.float64)
    return out


def c_oracle(model, data, k, state, gm, regU=REG, regI=REG, regB=REG, regC=REGC):
    from oracle import oracle_c
    u, j, ctx, r = tuples_for(model, data)
    st = {n: np.array(a, dtype=np.float64, copy=True) for n, a in state.items()}
    return oracle_c.Oracle(model, k, data.n_users, data.n_items, data.n_conds, u, j, ctx, r, data.ctx_ptr,
                           data.ctx_conds, st, gm, regU, regI, regB, regC)


class OracleEngine:
    """The CPU oracle behind tests.hostmirror.recommender's engine interface (tests only: the product engine is
    recommender.GpuEngine)."""

    def __init__(self, model, k, data, tuples, hp, flags=0, device=0):
        from oracle import oracle_c
        self.model, self.k, self.data, self.tuples, self.hp = model, k, data, tuples, hp
        self.oracle_c = oracle_c
        self.orc = None

    def set_states(self, st):
        u, j, ctx, r = self.tuples
        st = {n: np.array(a, dtype=np.float64, copy=True) for n, a in st.items()}
        d = self.data
        if self.model in self.oracle_c.SIM_MODEL_IDS:   # SVD++ / CAMF_ICS / CAMF_LCS / CAMF_MCS: carskit_oracle_sim.c
            self.orc = self.oracle_c.SimOracle(self.model, self.k, d.n_users, d.n_items, d.n_conds, u, j, ctx, r, d.ctx_ptr, d.ctx_conds,
                                               d.empty_conds, st, self.hp["gm"], self.hp["regU"], self.hp["regI"], self.hp["regB"],
                                               self.hp["regC"], n_ctx_dims=max(1, d.n_dims))
            return
        self.orc = self.oracle_c.Oracle(self.model, self.k, d.n_users, d.n_items, d.n_conds, u, j,
                                        ctx if ctx is not None else np.zeros(len(r), np.int32), r, d.ctx_ptr,
                                        d.ctx_conds, st, self.hp["gm"], self.hp["regU"], self.hp["regI"],
                                        self.hp["regB"], self.hp["regC"])

    def get_states(self):
        return {n: a for n, a in self.orc.state.items() if a is not None}

    def epoch(self, lr):
        return self.orc.epoch(lr)

    def eval_ratings(self, u, j, ctx, r, lo, hi):
        if self.model in self.oracle_c.SIM_MODEL_IDS:    # Recommender.evalRatings (Recommender.java:504-594) over predict()
            pred = np.array([min(max(self.orc.predict(int(a), int(b), -1 if ctx is None else int(c)), lo), hi)
                             for a, b, c in zip(u, j, ctx if ctx is not None else u)])
            err = np.abs(np.asarray(r) - pred)
            rerr = np.abs(np.asarray(r) - np.floor(pred / lo + 0.5) * lo)
            n = len(err)
            mae = float(err.sum() / n)
            return {"MAE": mae, "RMSE": float(np.sqrt((err * err).sum() / n)), "NMAE": mae / (hi - lo), "rMAE": float(rerr.sum() / n),
                    "rRMSE": float(np.sqrt((rerr * rerr).sum() / n)), "n": n}
        return self.orc.eval_ratings(u, j, ctx, r, lo, hi)

    def eval_rankings(self, train, test, bin_thold, num_recs, num_ignore, strategy):
        from oracle import rank_oracle
        tup = lambda t: list(zip(*(np.asarray(a).tolist() for a in t)))
        res, _ = rank_oracle.eval_rankings(lambda u, j, c: self.orc.predict(u, j, c), tup(train), tup(test), bin_thold,
                                           num_recs, strategy, num_ignore)
        return res

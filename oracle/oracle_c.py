"""ctypes loader for oracle/libcarskit_oracle.so.  TEST INFRASTRUCTURE ONLY: importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from carskit_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODEL_IDS = {"BiasedMF": 0, "CAMF_C": 1, "CAMF_CI": 2, "CAMF_CU": 3, "CAMF_CUCI": 4, "PMF": 5}


class OrcProblem(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("k", C.c_int32),
        ("n_users", C.c_int32), ("n_items", C.c_int32), ("n_conds", C.c_int32),
        ("n", C.c_int64),
        ("u", C.c_void_p), ("j", C.c_void_p), ("ctx", C.c_void_p), ("r", C.c_void_p),
        ("ctx_ptr", C.c_void_p), ("ctx_conds", C.c_void_p),
        ("P", C.c_void_p), ("Q", C.c_void_p), ("userBias", C.c_void_p), ("itemBias", C.c_void_p),
        ("condBias", C.c_void_p), ("ucBias", C.c_void_p), ("icBias", C.c_void_p),
        ("globalMean", C.c_double),
        ("regU", C.c_double), ("regI", C.c_double), ("regB", C.c_double), ("regC", C.c_double),
    ]


class OrcSchedule(C.Structure):
    _fields_ = [
        ("lRate", C.c_double), ("maxLRate", C.c_double), ("decay", C.c_double),
        ("boldDriver", C.c_int32), ("earlyStop", C.c_int32),
        ("loss", C.c_double), ("last_loss", C.c_double),
        ("measure", C.c_double), ("last_measure", C.c_double),
    ]


class OrcFM(C.Structure):
    _fields_ = [
        ("k", C.c_int32), ("n_users", C.c_int32), ("n_items", C.c_int32), ("n_conds", C.c_int32),
        ("n_ctx_dims", C.c_int32), ("p", C.c_int32), ("size", C.c_int64),
        ("u", C.c_void_p), ("j", C.c_void_p), ("ctx", C.c_void_p), ("r", C.c_void_p),
        ("w0", C.c_double), ("w", C.c_void_p), ("V", C.c_void_p), ("Q", C.c_void_p), ("errors", C.c_void_p),
        ("regLw", C.c_double), ("regLf", C.c_double),
    ]


class OrcSim(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("k", C.c_int32), ("n_users", C.c_int32), ("n_items", C.c_int32), ("n_conds", C.c_int32),
        ("numF", C.c_int32), ("n", C.c_int64),
        ("u", C.c_void_p), ("j", C.c_void_p), ("ctx", C.c_void_p), ("r", C.c_void_p),
        ("ctx_ptr", C.c_void_p), ("ctx_conds", C.c_void_p), ("empty_conds", C.c_void_p),
        ("ui_ptr", C.c_void_p), ("ui_items", C.c_void_p),
        ("P", C.c_void_p), ("Q", C.c_void_p), ("userBias", C.c_void_p), ("itemBias", C.c_void_p), ("Y", C.c_void_p),
        ("ccMatrix", C.c_void_p), ("cfMatrix", C.c_void_p), ("cVector", C.c_void_p),
        ("globalMean", C.c_double), ("regU", C.c_double), ("regI", C.c_double), ("regB", C.c_double), ("regC", C.c_double),
        ("upbound", C.c_double), ("lowbound", C.c_double),
    ]


class OrcJRandom(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("nextNextGaussian", C.c_double), ("haveNextNextGaussian", C.c_int32)]


def build(force=False):
    so = os.path.join(_HERE, "libcarskit_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("carskit_oracle.c", "carskit_oracle_fm.c", "carskit_oracle_sim.c", "carskit_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_sgd_epoch.restype = C.c_double
        L.orc_sgd_epoch.argtypes = [C.POINTER(OrcProblem), C.c_double]
        L.orc_is_converged.restype = C.c_int
        L.orc_is_converged.argtypes = [C.POINTER(OrcSchedule), C.c_int, C.c_int]
        L.orc_build_model.restype = C.c_int
        L.orc_build_model.argtypes = [C.POINTER(OrcProblem), C.POINTER(OrcSchedule), C.c_int, C.c_void_p, C.c_void_p]
        L.orc_predict_items.restype = None
        L.orc_predict_items.argtypes = [C.POINTER(OrcProblem), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.orc_predict.restype = C.c_double
        L.orc_predict.argtypes = [C.POINTER(OrcProblem), C.c_int32, C.c_int32, C.c_int32]
        L.orc_eval_ratings.restype = C.c_int64
        L.orc_eval_ratings.argtypes = [C.POINTER(OrcProblem), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_global_mean.restype = C.c_double
        L.orc_global_mean.argtypes = [C.c_void_p, C.c_int64]
        L.orc_jrandom_seed.argtypes = [C.POINTER(OrcJRandom), C.c_int64]
        L.orc_jrandom_next.restype = C.c_int32
        L.orc_jrandom_next.argtypes = [C.POINTER(OrcJRandom), C.c_int]
        L.orc_jrandom_next_double.restype = C.c_double
        L.orc_jrandom_next_double.argtypes = [C.POINTER(OrcJRandom)]
        L.orc_jrandom_next_gaussian.restype = C.c_double
        L.orc_jrandom_next_gaussian.argtypes = [C.POINTER(OrcJRandom)]
        L.orc_init_gaussian.argtypes = [C.POINTER(OrcJRandom), C.c_void_p, C.c_int64, C.c_double, C.c_double]
        L.orc_init_uniform.argtypes = [C.POINTER(OrcJRandom), C.c_void_p, C.c_int64, C.c_double]
        L.orc_sim_predict.restype = C.c_double
        L.orc_sim_predict.argtypes = [C.POINTER(OrcSim), C.c_int32, C.c_int32, C.c_int32]
        L.orc_sim_epoch.restype = C.c_double
        L.orc_sim_epoch.argtypes = [C.POINTER(OrcSim), C.c_double]
        L.orc_fm_predict.restype = C.c_double
        L.orc_fm_predict.argtypes = [C.POINTER(OrcFM), C.c_int32, C.c_int32, C.c_int32]
        L.orc_fm_init.argtypes = [C.POINTER(OrcFM)]
        L.orc_fm_sweep.restype = C.c_double
        L.orc_fm_sweep.argtypes = [C.POINTER(OrcFM)]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


STATE_NAMES = ("P", "Q", "userBias", "itemBias", "condBias", "ucBias", "icBias")


class Oracle:
    """One recommender instance over flat numpy arrays (state arrays are updated in place).

    `state` maps names in STATE_NAMES to float64 arrays; tuples are int32 u,j,ctx and float64 r in
    the reference's iteration order."""

    def __init__(self, model, k, n_users, n_items, n_conds, u, j, ctx, r, ctx_ptr, ctx_conds, state,
                 global_mean, regU, regI, regB, regC):
        self.L = lib()
        c32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        self.u, self.j, self.ctx = c32(u), c32(j), c32(ctx)
        self.r = np.ascontiguousarray(r, dtype=np.float64)
        self.ctx_ptr, self.ctx_conds = c32(ctx_ptr), c32(ctx_conds)
        self.state = {}
        for name in STATE_NAMES:
            a = state.get(name)
            if a is not None:
                a = np.ascontiguousarray(a, dtype=np.float64)
            self.state[name] = a
        mid = MODEL_IDS[model] if isinstance(model, str) else int(model)
        self.p = OrcProblem(mid, k, n_users, n_items, n_conds, len(self.r), _p(self.u), _p(self.j), _p(self.ctx),
                            _p(self.r), _p(self.ctx_ptr), _p(self.ctx_conds),
                            *[_p(self.state[nm]) for nm in STATE_NAMES],
                            global_mean, regU, regI, regB, regC)

    def epoch(self, lrate):
        return self.L.orc_sgd_epoch(C.byref(self.p), lrate)

    def build_model(self, num_iters, init_lrate, max_lrate=-1.0, bold_driver=False, decay=-1.0, early_stop=0):
        s = OrcSchedule(init_lrate, max_lrate, decay, int(bold_driver), early_stop, 0, 0, 0, 0)
        losses = np.zeros(num_iters)
        lrs = np.zeros(num_iters)
        n = self.L.orc_build_model(C.byref(self.p), C.byref(s), num_iters, _p(losses), _p(lrs))
        return losses[:n], lrs[:n], s

    def predict(self, u, j, ctx):
        return self.L.orc_predict(C.byref(self.p), u, j, ctx)

    def predict_items(self, u, ctx, items):
        items = np.ascontiguousarray(items, dtype=np.int32)
        out = np.empty(len(items))
        self.L.orc_predict_items(C.byref(self.p), u, ctx, len(items), _p(items), _p(out))
        return out

    def eval_ratings(self, tu, tj, tctx, tr, min_rate, max_rate, want_preds=False):
        tu = np.ascontiguousarray(tu, dtype=np.int32)
        tj = np.ascontiguousarray(tj, dtype=np.int32)
        tctx = None if tctx is None else np.ascontiguousarray(tctx, dtype=np.int32)
        tr = np.ascontiguousarray(tr, dtype=np.float64)
        out = np.zeros(5)
        preds = np.zeros(len(tr)) if want_preds else None
        n = self.L.orc_eval_ratings(C.byref(self.p), len(tr), _p(tu), _p(tj), _p(tctx), _p(tr), min_rate, max_rate,
                                    _p(out), _p(preds))
        res = dict(zip(("MAE", "RMSE", "NMAE", "rMAE", "rRMSE"), out.tolist()))
        res["n"] = n
        return (res, preds) if want_preds else res


def global_mean(r):
    r = np.ascontiguousarray(r, dtype=np.float64)
    return lib().orc_global_mean(_p(r), len(r))


class JRandom:
    def __init__(self, seed):
        self.L = lib()
        self.g = OrcJRandom()
        self.L.orc_jrandom_seed(C.byref(self.g), seed)

    def next_int(self):
        return self.L.orc_jrandom_next(C.byref(self.g), 32)

    def next_double(self):
        return self.L.orc_jrandom_next_double(C.byref(self.g))

    def next_gaussian(self):
        return self.L.orc_jrandom_next_gaussian(C.byref(self.g))

    def gaussian(self, shape, mean=0.0, sigma=0.1):
        a = np.zeros(shape)
        self.L.orc_init_gaussian(C.byref(self.g), _p(a), a.size, mean, sigma)
        return a

    def uniform(self, shape, rng=1.0):
        a = np.zeros(shape)
        self.L.orc_init_uniform(C.byref(self.g), _p(a), a.size, rng)
        return a


class FMOracle:
    """The reference's FM over flat arrays (w, V are updated in place; w0 lives in self.m.w0)."""

    def __init__(self, k, n_users, n_items, n_conds, n_ctx_dims, u, j, ctx, r, w0, w, V, regLw, regLf):
        self.L = lib()
        self.u = np.ascontiguousarray(u, dtype=np.int32)
        self.j = np.ascontiguousarray(j, dtype=np.int32)
        self.ctx = np.ascontiguousarray(ctx, dtype=np.int32)
        self.r = np.ascontiguousarray(r, dtype=np.float64)
        p = n_users + n_items + n_conds
        self.w = np.array(w, dtype=np.float64).reshape(p)
        self.V = np.array(V, dtype=np.float64).reshape(p, k)
        n = len(self.r)
        self.Q = np.zeros((n, k))
        self.errors = np.zeros(n)
        self.m = OrcFM(k, n_users, n_items, n_conds, n_ctx_dims, p, n, _p(self.u), _p(self.j), _p(self.ctx),
                       _p(self.r), w0, _p(self.w), _p(self.V), _p(self.Q), _p(self.errors), regLw, regLf)

    @property
    def w0(self):
        return self.m.w0

    def init(self):
        self.L.orc_fm_init(C.byref(self.m))

    def sweep(self):
        return self.L.orc_fm_sweep(C.byref(self.m))

    def predict(self, u, j, c):
        return self.L.orc_fm_predict(C.byref(self.m), u, j, c)


SIM_MODEL_IDS = {"SVD++": 6, "CAMF_ICS": 7, "CAMF_LCS": 8, "CAMF_MCS": 9}
SIM_STATE_NAMES = ("P", "Q", "userBias", "itemBias", "Y", "ccMatrix", "cfMatrix", "cVector")


def user_items_csr(u, j, n_users):
    """librec SparseMatrix.rowColumnsCache of the 2-D train matrix: the items of every user, ascending (SVDPlusPlus.java:52)."""
    u = np.asarray(u, dtype=np.int64)
    order = np.lexsort((np.asarray(j), u))
    ptr = np.zeros(n_users + 1, dtype=np.int32)
    np.add.at(ptr, u + 1, 1)
    return np.cumsum(ptr).astype(np.int32), np.asarray(j, dtype=np.int32)[order]


class SimOracle:
    """SVD++ / CAMF_ICS / CAMF_LCS / CAMF_MCS over flat numpy arrays (state updated in place)."""

    def __init__(self, model, k, n_users, n_items, n_conds, u, j, ctx, r, ctx_ptr, ctx_conds, empty_conds, state, global_mean,
                 regU, regI, regB, regC, n_ctx_dims=1):
        self.L = lib()
        c32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        self.u, self.j, self.ctx = c32(u), c32(j), c32(ctx)
        self.r = np.ascontiguousarray(r, dtype=np.float64)
        self.ctx_ptr, self.ctx_conds, self.empty = c32(ctx_ptr), c32(ctx_conds), c32(empty_conds)
        self.state = {nm: (None if state.get(nm) is None else np.ascontiguousarray(state[nm], dtype=np.float64)) for nm in SIM_STATE_NAMES}
        self.ui_ptr = self.ui_items = None
        if model == "SVD++":
            self.ui_ptr, self.ui_items = user_items_csr(self.u, self.j, n_users)
        numF = self.state["cfMatrix"].shape[1] if self.state["cfMatrix"] is not None else 0
        self.p = OrcSim(SIM_MODEL_IDS[model], k, n_users, n_items, n_conds, numF, len(self.r), _p(self.u), _p(self.j), _p(self.ctx),
                        _p(self.r), _p(self.ctx_ptr), _p(self.ctx_conds), _p(self.empty), _p(self.ui_ptr), _p(self.ui_items),
                        *[_p(self.state[nm]) for nm in SIM_STATE_NAMES], global_mean, regU, regI, regB, regC,
                        1.0 / np.sqrt(n_ctx_dims), 1.0 / (10.0 ** 100))

    def epoch(self, lrate):
        return self.L.orc_sim_epoch(C.byref(self.p), lrate)

    def predict(self, u, j, ctx=-1):
        return self.L.orc_sim_predict(C.byref(self.p), u, j, ctx)

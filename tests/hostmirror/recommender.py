"""Host-side mirror of the reference's recommender API for the accelerated path -- same class names, hook names
and lifecycle as carskit.generic.{Recommender, IterativeRecommender, ContextRecommender} and the model classes
(reference files cited per class), so parity tests and the driver read like the reference:

    algo = CAMF_CI(train, test, fold, conf); algo.execute(); algo.measures["RMSE"]

execute() = initModel() -> buildModel() -> evalRatings()  (Recommender.java:319-366).  buildModel() hands the
tuples and the initial model to the compute engine and keeps the reference's control flow: one engine epoch per
iteration, then isConverged()/updateLRate() on the host (IterativeRecommender.java:145-229).

The engine is libcarskit_mi355x.so (GpuEngine over carskit_amd.capi); there is no CPU engine in this package --
the test-suite injects one (the oracle) through `engine_factory` to exercise this plumbing without a GPU.
"""
import math
import time

import numpy as np

from carskit_amd import capi, synth
from .config import java_float


def _f32(x):
    return float(np.float32(x))


class Conf:
    """The static hyper-parameters the reference parses once per run (Recommender.java:194-247,
    IterativeRecommender.java:80-103, FM.java:53-54), as the doubles the Java floats promote to."""

    def __init__(self, cf=None, **over):
        self.num_factors, self.num_iters = 10, 100
        self.init_lrate, self.max_lrate, self.bold_driver, self.decay = java_float("0.01"), -1.0, False, -1.0
        self.reg = self.regU = self.regI = self.regB = self.regC = java_float("0.01")
        self.early_stop, self.verbose = None, True
        self.reg_lw = self.reg_lf = 0.0
        self.num_f = 10     # `-f` of the recommender line (CAMF_LCS.java:37)
        self.init_mean, self.init_std = 0.0, 0.1
        self.init_seed = 1
        self.flags = 0
        # item.ranking / ratings.setup / eval.strategy (Recommender.java:211-217,242; CARSKit.java:262)
        self.is_ranking, self.num_recs, self.num_ignore, self.bin_thold, self.eval_strategy = False, 10, -1, -1.0, "ucu"
        if cf is not None:
            lc = cf.get_param_options("learn.rate")
            if lc is not None:
                self.init_lrate = java_float(lc.get_main_param())
                self.max_lrate = lc.get_float("-max", -1.0)
                self.bold_driver = lc.contains("-bold-driver")
                self.decay = lc.get_float("-decay", -1.0)
            ro = cf.get_param_options("reg.lambda")
            if ro is not None:
                self.reg = java_float(ro.get_main_param())
                self.regU, self.regI = ro.get_float("-u", self.reg), ro.get_float("-i", self.reg)
                self.regB, self.regC = ro.get_float("-b", self.reg), ro.get_float("-c", self.reg)
            self.num_factors = cf.get_int("num.factors", 10)
            self.num_iters = cf.get_int("num.max.iter", 100)
            ev = cf.get_param_options("evaluation.setup")
            if ev is not None:
                es = ev.get_string("--early-stop")
                if es is not None:
                    self.early_stop = {"loss": "Loss", "mae": "MAE", "rmse": "RMSE"}.get(es.lower())
                self.init_seed = ev.get_long("--rand-seed", 1)
            out = cf.get_param_options("output.setup")
            if out is not None:
                self.verbose = out.is_on("-verbose", True)
            rk = cf.get_param_options("item.ranking")
            if rk is not None:
                self.is_ranking = rk.is_main_on()
                self.num_recs = rk.get_int("-topN", -1)
                if self.num_recs < 0:
                    self.num_recs = 10
                self.num_ignore = rk.get_int("-ignore", -1)
                if rk.contains("-diverse"):
                    raise ValueError("item.ranking -diverse (item-similarity diversity) is not on the accelerated path")
            rs = cf.get_param_options("ratings.setup")
            if rs is not None:
                self.bin_thold = rs.get_float("-threshold", -1.0)   # a Java float, promoted where it is compared
            self.eval_strategy = (cf.get_string("eval.strategy") or "ucu").lower()
            rec = cf.get_param_options("recommender")
            if rec is not None:
                self.num_f = rec.get_int("-f", 10)
            fm = cf.get_param_options("FM")
            if fm is not None:
                self.reg_lw, self.reg_lf = fm.get_float("-lw", 0.0), fm.get_float("-lf", 0.0)
        for k, v in over.items():
            setattr(self, k, v)


class GpuEngine:
    """One recommender instance on the GPU through the C ABI."""

    def __init__(self, model, k, data, tuples, hp, flags=0, device=0):
        u, j, ctx, r = tuples
        if model in ("CAMF_C", "SVD++", "CAMF_ICS", "CAMF_LCS", "CAMF_MCS"):   # one dependent chain in CRS order
            flags |= capi.FLAG_SCHED_SERIAL
        self.inst = capi.Instance(model, k, data.n_users, data.n_items, data.n_conds, device=device, flags=flags)
        self.inst.set_hparams(hp["regU"], hp["regI"], hp["regB"], hp["regC"], hp["gm"])
        if model in ("CAMF_ICS", "CAMF_LCS", "CAMF_MCS"):
            if data.empty_conds is None:
                raise ValueError("%s needs EmptyContextConditions (the ':na' condition of every dimension)" % model)
            self.inst.set_sim_params(hp.get("numF", 10), max(1, data.n_dims), data.empty_conds)
        if model in ("BiasedMF", "PMF", "SVD++"):
            self.inst.set_ratings(u, j, None, r)
        else:
            self.inst.set_ratings(u, j, ctx, r, data.ctx_ptr, data.ctx_conds)

    def set_states(self, st):
        self.inst.set_states(st)

    def get_states(self):
        return self.inst.get_states()

    def epoch(self, lr):
        return self.inst.train_epoch(lr)

    def eval_ratings(self, u, j, ctx, r, lo, hi):
        return self.inst.eval_ratings(u, j, ctx, r, lo, hi)

    def predict(self, u, j, ctx, bound=None):
        return self.inst.predict(u, j, ctx, bound)

    def eval_rankings(self, train, test, bin_thold, num_recs, num_ignore, strategy):
        return self.inst.eval_rankings(train, test, bin_thold, num_recs, num_ignore, strategy)

    def set_eval_ratings(self, u, j, ctx, r):      # test tuples resident on the device (per-epoch early-stop evaluation)
        self.inst.set_eval_ratings(u, j, ctx, r)
        self.eval_resident_ready = len(r) > 0      # an empty test set has nothing resident: fall back to the per-call path

    def eval_resident(self, lo, hi):
        return self.inst.eval_resident(lo, hi)


class Recommender:
    """carskit.generic.Recommender (src/carskit/generic/Recommender.java)."""

    algo_name = None
    is_cars = True

    def __init__(self, train, test, fold=-1, conf=None, engine_factory=None, log=None):
        self.trainMatrix, self.testMatrix, self.fold = train, test, fold
        self.conf = conf or Conf()
        self.foldInfo = " fold [%d]" % fold if fold > 0 else ""
        self.numUsers, self.numItems, self.numConditions = train.n_users, train.n_items, train.n_conds
        self.minRate, self.maxRate = train.min_rate, train.max_rate      # full data's rating scale (:198-200)
        self.globalMean = float(np.sum(train.r) / np.count_nonzero(train.r)) if train.n else float("nan")  # :265
        self.engine_factory = engine_factory or GpuEngine
        self.device = 0                                               # set by the driver: fold -> GPU round robin
        self.measures = {}
        self.log = log or (lambda *a: None)
        self.engine = None

    # hooks
    def initModel(self):
        raise NotImplementedError

    def buildModel(self):
        raise NotImplementedError

    def evalRatings(self):
        t = self.testMatrix
        if getattr(self.engine, "eval_resident_ready", False):
            res = self.engine.eval_resident(self.minRate, self.maxRate)
        else:
            tu, tj, tc, tr = self.test_tuples()
            res = self.engine.eval_ratings(tu, tj, tc, tr, self.minRate, self.maxRate)
        res["MPE"] = 0.0                                           # numPEs is never incremented (:569)
        return res

    def evalRankings(self):                                        # Recommender.java:668-964
        c, tr, te = self.conf, self.trainMatrix, self.testMatrix
        if c.num_recs < 1:
            raise ValueError("item.ranking -topN 0 (unbounded lists with a cut-off of 0) is not supported")
        res = self.engine.eval_rankings((tr.u, tr.j, tr.ctx, tr.r), (te.u, te.j, te.ctx, te.r), c.bin_thold, c.num_recs,
                                        c.num_ignore, "uc" if c.eval_strategy == "uc" else "ucu")
        res.pop("n_queries", None)
        return res

    def test_tuples(self):
        t = self.testMatrix
        return t.u, t.j, (t.ctx if self.is_cars else None), t.r

    def execute(self):                                              # Recommender.java:319-366
        t0 = time.time()
        self.initModel()
        self.buildModel()
        t1 = time.time()
        self.measures = self.evalRankings() if self.conf.is_ranking else self.evalRatings()   # :346
        t2 = time.time()
        self.measures["TrainTime"] = (t1 - t0) * 1e3
        self.measures["TestTime"] = (t2 - t1) * 1e3
        return self.measures


class IterativeRecommender(Recommender):
    """carskit.generic.IterativeRecommender (src/carskit/generic/IterativeRecommender.java)."""

    states = ()

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        c = self.conf
        self.numFactors, self.numIters = c.num_factors, c.num_iters
        self.lRate = c.init_lrate                                   # :106
        self.loss = self.last_loss = 0.0
        self.measure = self.last_measure = 0.0
        self.state = {}
        self.losses, self.lrates = [], []

    def train_tuples(self):
        d = self.trainMatrix
        if not self.is_cars:                                        # Recommender.initModel (:1076-1081): 2-D `train`
            u, j, r = synth.to_2d(d)
            return u, j, None, r
        return d.u, d.j, d.ctx, d.r

    def initModel(self):
        """P, Q ~ N(0, 0.1) row-major, then the model's bias containers in source order (IterativeRecommender.java:
        232-247 + the model class).  The reference draws them from an UNSEEDED static java.util.Random:
        no two reference runs agree, so any seeded stream is as faithful as any other; callers that need parity
        assign self.state themselves before buildModel()."""
        if not self.state:
            self.trainMatrix.meta["num_f"] = self.conf.num_f
            self.state = synth.init_state(self.algo_name, self.trainMatrix, self.numFactors, seed=self.conf.init_seed)

    def buildModel(self):
        hp = {"regU": self.conf.regU, "regI": self.conf.regI, "regB": self.conf.regB, "regC": self.conf.regC,
              "gm": self.globalMean, "numF": self.conf.num_f}
        self.engine = self.engine_factory(self.algo_name, self.numFactors, self.trainMatrix, self.train_tuples(), hp,
                                          flags=self.conf.flags, device=self.device)
        self.engine.set_states(self.state)                          # copy-in
        if self.conf.early_stop in ("MAE", "RMSE") and hasattr(self.engine, "set_eval_ratings"):
            self.engine.set_eval_ratings(*self.test_tuples())       # evaluated after every epoch: keep it on the device
        for it in range(1, self.numIters + 1):
            self.lrates.append(self.lRate)
            self.loss = self.engine.epoch(self.lRate)               # the for(MatrixEntry me : trainMatrix) body
            self.losses.append(self.loss)
            if self.isConverged(it):
                break
        self.state = self.engine.get_states()                       # copy-back

    def isConverged(self, it):                                      # IterativeRecommender.java:145-199
        delta_loss = _f32(self.last_loss - self.loss)
        es = self.conf.early_stop
        if es == "Loss":
            self.measure, self.last_measure = self.loss, self.last_loss
        elif es in ("MAE", "RMSE"):
            self.measure = self.evalRatings()[es]
        delta_measure = _f32(self.last_measure - self.measure)
        if self.conf.verbose:
            self.log("%s%s iter %d: loss = %s, delta_loss = %s, learn_rate = %s" % (
                self.algo_name, self.foldInfo, it, _f32(self.loss), delta_loss, _f32(self.lRate)))
        if math.isnan(self.loss) or math.isinf(self.loss):
            raise FloatingPointError("Loss = NaN or Infinity: current settings does not fit the recommender! "
                                     "Change the settings and try again!")
        converged = abs(self.loss) < 1e-5 or (0 < delta_measure < 1e-5)
        if not converged:
            self.updateLRate(it)
        self.last_loss, self.last_measure = self.loss, self.measure
        return converged

    def updateLRate(self, it):                                      # IterativeRecommender.java:216-229
        if self.lRate <= 0:
            return
        c = self.conf
        if c.bold_driver and it > 1:
            self.lRate = self.lRate * 1.05 if abs(self.last_loss) > abs(self.loss) else self.lRate * 0.5
        elif 0 < c.decay < 1:
            self.lRate *= c.decay
        if c.max_lrate > 0 and self.lRate > c.max_lrate:
            self.lRate = c.max_lrate

    def predict(self, u, j, c=None, bound=False):
        return self.engine.predict(u, j, c, (self.minRate, self.maxRate) if bound else None)


class ContextRecommender(IterativeRecommender):
    """carskit.generic.ContextRecommender (src/carskit/generic/ContextRecommender.java)."""
    is_cars = True


class BiasedMF(IterativeRecommender):   # src/carskit/alg/baseline/cf/BiasedMF.java
    algo_name = "BiasedMF"
    is_cars = False


class PMF(IterativeRecommender):        # src/carskit/alg/baseline/cf/PMF.java
    algo_name = "PMF"
    is_cars = False


class CAMF_C(ContextRecommender):       # src/carskit/alg/cars/adaptation/dependent/dev/CAMF_C.java
    algo_name = "CAMF_C"


class CAMF_CI(ContextRecommender):      # .../dev/CAMF_CI.java
    algo_name = "CAMF_CI"


class CAMF_CU(ContextRecommender):      # .../dev/CAMF_CU.java
    algo_name = "CAMF_CU"


class CAMF_CUCI(ContextRecommender):    # .../dev/CAMF_CUCI.java
    algo_name = "CAMF_CUCI"


class SVDPlusPlus(IterativeRecommender):   # src/carskit/alg/baseline/cf/SVDPlusPlus.java (2-D train matrix)
    algo_name = "SVD++"
    is_cars = False


class _SimCAMF(ContextRecommender):
    """The similarity-based CAMF recommenders are top-N models: their constructors set isRankingPred = true
    (CAMF_ICS.java:31, CAMF_LCS.java:31, CAMF_MCS.java:37), so execute() evaluates with evalRankings()."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        import copy
        self.conf = copy.copy(self.conf)
        self.conf.is_ranking = True


class CAMF_ICS(_SimCAMF):               # src/carskit/alg/cars/adaptation/dependent/sim/CAMF_ICS.java
    algo_name = "CAMF_ICS"


class CAMF_LCS(_SimCAMF):               # .../sim/CAMF_LCS.java
    algo_name = "CAMF_LCS"


class CAMF_MCS(_SimCAMF):               # .../sim/CAMF_MCS.java
    algo_name = "CAMF_MCS"


class FM(ContextRecommender):
    """src/carskit/alg/cars/adaptation/dependent/FM.java: w0 = 0, w ~ U(0,1), V ~ N(0, 0.1); numIters ALS sweeps,
    no convergence check."""
    algo_name = "FM"

    def initModel(self):
        if not self.state:
            rng = np.random.default_rng(self.conf.init_seed)
            p = self.numUsers + self.numItems + self.numConditions
            self.state = {"w0": 0.0, "w": rng.random(p), "V": 0.1 * rng.standard_normal((p, self.numFactors))}

    def buildModel(self):
        d = self.trainMatrix
        self.engine = capi.FMInstance(self.numFactors, d.n_users, d.n_items, d.n_conds, max(1, d.n_dims), device=self.device)
        self.engine.set_hparams(self.conf.reg_lw, self.conf.reg_lf)
        self.engine.set_ratings(d.u, d.j, d.ctx, d.r)
        self.engine.set_model(self.state["w0"], self.state["w"], self.state["V"])
        self.engine.train(self.numIters)
        w0, w, V = self.engine.get_model()
        self.state = {"w0": w0, "w": w, "V": V}

    def evalRankings(self):                                        # Recommender.java:668-964 with FM.predict
        c, tr, te = self.conf, self.trainMatrix, self.testMatrix
        if c.num_recs < 1:
            raise ValueError("item.ranking -topN 0 (unbounded lists with a cut-off of 0) is not supported")
        res = self.engine.eval_rankings((tr.u, tr.j, tr.ctx, tr.r), (te.u, te.j, te.ctx, te.r), c.bin_thold, c.num_recs,
                                        c.num_ignore, "uc" if c.eval_strategy == "uc" else "ucu")
        res.pop("n_queries", None)
        return res

    def evalRatings(self):
        t = self.testMatrix
        pred = self.engine.predict(t.u, t.j, t.ctx, bound=(self.minRate, self.maxRate))
        ok = ~np.isnan(pred)
        err = np.abs(t.r[ok] - pred[ok])
        rpred = np.floor(pred[ok] / self.minRate + 0.5) * self.minRate
        rerr = np.abs(t.r[ok] - rpred)
        n = int(ok.sum())
        mae = float(err.sum() / n)
        return {"MAE": mae, "RMSE": float(np.sqrt((err * err).sum() / n)), "NMAE": mae / (self.maxRate - self.minRate),
                "rMAE": float(rerr.sum() / n), "rRMSE": float(np.sqrt((rerr * rerr).sum() / n)), "MPE": 0.0, "n": n}


# the reference's factory switch (src/carskit/main/CARSKit.java:461-469,700-712,742), lower-cased names
RECOMMENDERS = {"biasedmf": BiasedMF, "pmf": PMF, "svd++": SVDPlusPlus, "camf_c": CAMF_C, "camf_ci": CAMF_CI, "camf_cu": CAMF_CU,
                "camf_cuci": CAMF_CUCI, "camf_ics": CAMF_ICS, "camf_lcs": CAMF_LCS, "camf_mcs": CAMF_MCS, "fm": FM}


def get_eval_info(ms, conf=None):
    """Recommender.getEvalInfo (Recommender.java:437-499): ranking branch with the reference's exact (irregular)
    separators, rating branch incl. its 'NAME' typo for NMAE."""
    if conf is not None and conf.is_ranking:
        n = conf.num_recs
        if n != 10:
            fmt = ("Pre5: %.6f,Pre10: %.6f, Pre{n}: %.6f, Rec5: %.6f, Rec10: %.6f, Rec{n}: %.6f, "
                   "AUC5: %.6f, AUC10: %.6f, AUC{n}: %.6f, MAP5: %.6f, MAP10: %.6f, MAP{n}: %.6f, "
                   "NDCG5: %.6f, NDCG10: %.6f,NDCG{n}: %.6f,MRR5: %.6f, MRR10: %.6f,MRR{n}: %.6f").format(n=n)
            keys = ("Pre5", "Pre10", "PreN", "Rec5", "Rec10", "RecN", "AUC5", "AUC10", "AUCN", "MAP5", "MAP10", "MAPN",
                    "NDCG5", "NDCG10", "NDCGN", "MRR5", "MRR10", "MRRN")
        else:
            fmt = ("Pre5: %.6f,Pre10: %.6f, Rec5: %.6f, Rec10: %.6f, AUC5: %.6f, AUC10: %.6f, MAP5: %.6f, MAP10: %.6f,"
                   "NDCG5: %.6f, NDCG10: %.6f,MRR5: %.6f, MRR10: %.6f")
            keys = ("Pre5", "Pre10", "Rec5", "Rec10", "AUC5", "AUC10", "MAP5", "MAP10", "NDCG5", "NDCG10", "MRR5", "MRR10")
        return fmt % tuple(ms[k] for k in keys)
    return "MAE: %.6f, RMSE: %.6f, NAME: %.6f, rMAE: %.6f, rRMSE: %.6f, MPE: %.6f" % (
        ms["MAE"], ms["RMSE"], ms["NMAE"], ms["rMAE"], ms["rRMSE"], ms.get("MPE", 0.0))

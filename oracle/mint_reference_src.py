#!/usr/bin/env python3
"""Mint tests/golden/reference_src_*.json by EXECUTING the reference's own Java source for the hot path (build container only).

    python oracle/mint_reference_src.py [/root/reference]

For every SGD recommender of the path -- BiasedMF, PMF, CAMF_C, CAMF_CI, CAMF_CU, CAMF_CUCI and SVD++, CAMF_ICS, CAMF_LCS,
CAMF_MCS -- the reference's
`buildModel()` is run, statement by statement, from the text of its .java files (src/carskit/alg/**, src/carskit/generic/**) by
oracle/jvm/javasrc.py: `predict(u, j, c, bound)`, the model's own `predict`, `getConditions`, `isConverged` and `updateLRate` are the
reference's methods too (resolved through the class chain like Java's virtual dispatch), and every `P.get / P.add / rowMult / userBias.add
/ for (MatrixEntry me : trainMatrix)` goes into the vendored librec jar's BYTECODE through oracle/jvm/interp.py.  Only what lies outside the
loop is provided by this script: the initial containers, the hyper-parameters as the Java fields would hold them (float fields as
floats), `rateDao`'s id maps, and guava's Table for CAMF_CUCI's bias tables.

Inputs and outputs (model state after the epochs, the loss and learning rate of every epoch) are written as data, doubles as hex --
tests/test_reference_src_golden.py then requires the C oracle, the Python restatement and (on a GPU) the strict fp64 kernels to
reproduce them BIT FOR BIT.  This is the closest thing to running the reference that a JVM-less image allows: its own statements, its own
operator order, its own reads-before-writes -- interpreted, not restated.  The reference's text is read where it lies; none of it enters
this repository.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.jvm import javasrc  # noqa: E402
from oracle.jvm.interp import VM, Box, GuavaMultimap, GuavaTable, JString, darray, dmatrix, f32, to_list  # noqa: E402

DM, DV, SM = "librec/data/DenseMatrix", "librec/data/DenseVector", "librec/data/SparseMatrix"
CLASS_MAP = {"DenseMatrix": DM, "DenseVector": DV, "SparseMatrix": SM, "SymmMatrix": "librec/data/SymmMatrix",
             "Randoms": "librec/util/Randoms", "Stats": "librec/util/Stats"}


def hx(v):
    return float(v).hex()


class RateDao:
    """carskit.data.processor.DataDAO, the three getters the loop uses (DataDAO.java:945-951, 1038-1046): id maps built by the caller"""

    def __init__(self, ui_user, ui_item, ctx_keys):
        self.ui_user, self.ui_item, self.ctx_keys = ui_user, ui_item, ctx_keys

    def jcall(self, vm, name, desc, args):
        a = [x.v if isinstance(x, Box) else x for x in args]
        if name == "getUserIdFromUI":
            return int(self.ui_user[a[0]])
        if name == "getItemIdFromUI":
            return int(self.ui_item[a[0]])
        if name == "getContextId":
            return JString(self.ctx_keys[a[0]])
        if name == "numContextDims":
            return len(self.ctx_keys[0].split(","))
        raise KeyError("rateDao." + name)


class BoxTable(GuavaTable):
    """guava Table<Integer,Integer,Double> as CAMF_CUCI uses it (get / put with auto-boxing)"""

    def jcall(self, vm, name, desc, args):
        if name == "get":
            return self.rows[args[0]][args[1]]
        return super().jcall(vm, name, desc, args)


def dense(vm, arr):
    m = vm.new_object(DM)
    vm.call(DM, "<init>", "([[D)V", [m, dmatrix(np.asarray(arr, dtype=np.float64))])
    return m


def vector(vm, arr):
    v = vm.new_object(DV)
    vm.call(DV, "<init>", "([D)V", [v, darray(np.asarray(arr, dtype=np.float64))])
    return v


def sparse(vm, n_rows, n_cols, cells):
    t, cm = GuavaTable(), GuavaMultimap()
    for r, c, v in cells:
        t.put(int(r), int(c), float(v))
        cm.put(int(c), int(r))
    s = vm.new_object(SM)
    vm.call(SM, "<init>", "(IILcom/google/common/collect/Table;Lcom/google/common/collect/Multimap;)V", [s, n_rows, n_cols, t, cm])
    return s


def problem(rng, n_users, n_items, n_dims, conds_per_dim, n_ratings):
    """a small contextual rating set in the reference's shapes: (ui pair, context) cells of a sparse matrix, CRS order"""
    n_conds = n_dims * conds_per_dim
    pairs, ctxs, cells = {}, {}, {}
    ui_user, ui_item, ctx_keys = [], [], []
    while len(cells) < n_ratings:
        u, j = int(rng.integers(n_users)), int(rng.integers(n_items))
        conds = tuple(d * conds_per_dim + int(rng.integers(conds_per_dim)) for d in range(n_dims))
        if (u, j) not in pairs:
            pairs[(u, j)] = len(pairs)
            ui_user.append(u)
            ui_item.append(j)
        key = ",".join(str(c) for c in conds)
        if key not in ctxs:
            ctxs[key] = len(ctxs)
            ctx_keys.append(key)
        cells[(pairs[(u, j)], ctxs[key])] = float(rng.integers(1, 6))
    order = sorted(cells)
    return {"n_users": n_users, "n_items": n_items, "n_conds": n_conds, "ui_user": ui_user, "ui_item": ui_item, "ctx_keys": ctx_keys,
            "cells": [[ui, c, cells[(ui, c)]] for ui, c in order]}


MODELS = {
    "BiasedMF": ("alg/baseline/cf/BiasedMF.java",), "PMF": ("alg/baseline/cf/PMF.java",),
    "CAMF_C": ("alg/cars/adaptation/dependent/dev/CAMF_C.java", "alg/cars/adaptation/dependent/CAMF.java", "generic/ContextRecommender.java"),
    "CAMF_CI": ("alg/cars/adaptation/dependent/dev/CAMF_CI.java", "alg/cars/adaptation/dependent/CAMF.java", "generic/ContextRecommender.java"),
    "CAMF_CU": ("alg/cars/adaptation/dependent/dev/CAMF_CU.java", "alg/cars/adaptation/dependent/CAMF.java", "generic/ContextRecommender.java"),
    "CAMF_CUCI": ("alg/cars/adaptation/dependent/dev/CAMF_CUCI.java", "generic/ContextRecommender.java"),
    # the remaining SGD recommenders of the family
    "SVD++": ("alg/baseline/cf/SVDPlusPlus.java", "alg/baseline/cf/BiasedMF.java"),
    "CAMF_ICS": ("alg/cars/adaptation/dependent/sim/CAMF_ICS.java", "alg/cars/adaptation/dependent/CAMF.java", "generic/ContextRecommender.java"),
    "CAMF_LCS": ("alg/cars/adaptation/dependent/sim/CAMF_LCS.java", "alg/cars/adaptation/dependent/CAMF.java", "generic/ContextRecommender.java"),
    "CAMF_MCS": ("alg/cars/adaptation/dependent/sim/CAMF_MCS.java", "alg/cars/adaptation/dependent/CAMF.java", "generic/ContextRecommender.java"),
}
STATE = {"BiasedMF": ("userBias", "itemBias"), "PMF": (), "CAMF_C": ("userBias", "itemBias", "condBias"), "CAMF_CI": ("userBias", "icBias"),
         "CAMF_CU": ("itemBias", "ucBias"), "CAMF_CUCI": ("ucBias", "icBias"),
         "SVD++": ("userBias", "itemBias", "Y"), "CAMF_ICS": ("ccMatrix",), "CAMF_LCS": ("cfMatrix",), "CAMF_MCS": ("cVector",)}
NUM_F = 4   # `-f` of CAMF_LCS (CAMF_LCS.java:37)
JAVA_FIELD = {"ccMatrix": "ccMatrix_ICS", "cfMatrix": "cfMatrix_LCS", "cVector": "cVector_MCS"}


class UserItemsCache:
    """train.rowColumnsCache(cacheSpec) (SVDPlusPlus.java:52): user -> the items of the user's row of the 2-D train matrix, ascending"""

    def __init__(self, cells2, n_users):
        self.items = [[] for _ in range(n_users)]
        for u, j, _ in cells2:
            self.items[u].append(j)

    def jcall(self, vm, name, desc, args):
        from oracle.jvm.interp import JCollection
        if name == "get":
            return JCollection([Box(j, "Integer") for j in self.items[args[0].v if isinstance(args[0], Box) else args[0]]])
        raise KeyError("userItemsCache." + name)


def source_dao(ref, vm, prob):
    """rateDao as a carskit.data.processor.DataDAO whose METHODS are interpreted from DataDAO.java (getUserCtxList, getItemList,
    getRatingCountByItem, getUserIdFromUI, getContextId ...); the maps readData() would have filled are built here from the problem"""
    dao = javasrc.This(vm, [os.path.join(ref, "src", "carskit", "data", "processor", "DataDAO.java")], {})
    ui_u, ui_i, ctx, u_rated, i_rated = javasrc.JHashMap(), javasrc.JHashMap(), javasrc.JBiMap(), javasrc.JHashMultimap(), javasrc.JHashMultimap()
    for ui, (u, j) in enumerate(zip(prob["ui_user"], prob["ui_item"])):
        ui_u.d[Box(ui, "Integer")] = Box(u, "Integer")
        ui_i.d[Box(ui, "Integer")] = Box(j, "Integer")
        u_rated.put(u, ui)
        i_rated.put(j, ui)
    for c, key in enumerate(prob["ctx_keys"]):
        ctx.d[JString(key)] = Box(c, "Integer")
    dim_ids = javasrc.JBiMap()
    for d in range(len(prob["ctx_keys"][0].split(","))):
        dim_ids.d[JString("dim%d" % d)] = Box(d, "Integer")
    dao.fields.update({"uiUserIds": ui_u, "uiItemIds": ui_i, "ctxIds": ctx, "idCtx": None, "uRatedList": u_rated, "iRatedList": i_rated,
                       "dimIds": dim_ids})
    return dao


def run_model(ref, model, prob, k, iters, seed, lrate=0.02, reg=1e-4, reg_c=1e-3, bold=True, test_cells=None, rank=None, drop_in=None,
              init_override=None, early_stop=None):
    """drop_in: (path of a *_GPU.java drop-in class of THIS repository, {simple class name: This / natives object}, make(vm) that fills
    that map) -- the drop-in is put in front of the reference's class chain, so its buildModel() override runs
    (oracle/check_java_binding.py); init_override: the initial containers of a minted case instead of fresh draws"""
    vm = VM([os.path.join(ref, "lib", "librec-v1.4-alpha.jar"), os.path.join(ref, "lib", "happy.coding.utils-1.2.6.jar")] if rank else
            os.path.join(ref, "lib", "librec-v1.4-alpha.jar"))
    rng = np.random.default_rng(seed)
    nu, ni, nc = prob["n_users"], prob["n_items"], prob["n_conds"]
    init = {"P": 0.1 * rng.standard_normal((nu, k)), "Q": 0.1 * rng.standard_normal((ni, k))}
    shapes = {"userBias": (nu,), "itemBias": (ni,), "condBias": (nc,), "ucBias": (nu, nc), "icBias": (ni, nc), "Y": (ni, k),
              "ccMatrix": (nc, nc), "cfMatrix": (nc, NUM_F), "cVector": (nc,)}
    n_dims = len(prob["ctx_keys"][0].split(","))
    upbound = float(1.0 / np.sqrt(float(n_dims)))     # CAMF_MCS.java:44
    for name in STATE[model]:
        if name == "ccMatrix":                        # symmetric, near 1 (CAMF_ICS.java:45-49 starts at exactly 1)
            a = 1.0 + 0.05 * rng.standard_normal(shapes[name])
            init[name] = (a + a.T) / 2
        elif name == "cfMatrix":
            init[name] = rng.random(shapes[name])
        elif name == "cVector":
            init[name] = rng.random(shapes[name]) * upbound
        else:
            init[name] = rng.random(shapes[name]) if name in ("ucBias", "icBias") and model != "CAMF_CUCI" else 0.1 * rng.standard_normal(shapes[name])
    if model in ("CAMF_ICS", "CAMF_LCS", "CAMF_MCS"):  # isRankingPred: P and Q start uniform (CAMF_ICS.java:36-41); any values do for the pin
        init["P"], init["Q"] = rng.random((nu, k)), rng.random((ni, k))
    src = [os.path.join(ref, "src", "carskit", p) for p in MODELS[model]] + \
          [os.path.join(ref, "src", "carskit", "generic", "IterativeRecommender.java"), os.path.join(ref, "src", "carskit", "generic", "Recommender.java")]
    if init_override:
        init = {n: np.array(a, dtype=np.float64).reshape(init[n].shape) for n, a in init_override.items()}
    if drop_in:
        src = [drop_in[0]] + src
        drop_in[2](vm)
    this = javasrc.This(vm, src, dict(CLASS_MAP, **(drop_in[1] if drop_in else {})))
    two_d = model in ("BiasedMF", "PMF", "SVD++")
    cells = prob["cells"]
    if two_d:   # DataDAO.toTraditionalSparseMatrix: users x items, the mean over contexts of every (user, item) pair
        acc = {}
        for ui, c, v in cells:
            key = (prob["ui_user"][ui], prob["ui_item"][ui])
            s, n = acc.get(key, (0.0, 0))
            acc[key] = (s + v, n + 1)
        cells2 = [[u, j, s / n] for (u, j), (s, n) in sorted(acc.items())]
        train2 = sparse(vm, nu, ni, cells2)
    train_ctx = sparse(vm, len(prob["ui_user"]), len(prob["ctx_keys"]), cells)
    vals = [v for _, _, v in cells]
    gm = 0.0
    for v in vals:
        gm += v
    gm = gm / sum(1 for v in vals if v != 0.0)                 # SparseMatrix.getGlobalAvg over the contextual train matrix (Recommender.java:265)
    F = this.fields
    F.update({"P": dense(vm, init["P"]), "Q": dense(vm, init["Q"]), "numFactors": k, "numIters": iters, "numUsers": nu, "numItems": ni,
              "numConditions": nc, "lRate": float(f32(lrate)), "initLRate": f32(lrate), "maxLRate": f32(-1.0), "decay": f32(-1.0),
              "isBoldDriver": bool(bold), "regU": f32(reg), "regI": f32(reg), "regB": f32(reg), "regC": f32(reg_c),
              "globalMean": gm, "loss": 0.0, "last_loss": 0.0, "measure": 0.0, "last_measure": 0.0, "earlyStopMeasure": None,
              "verbose": False, "isResultsOut": False, "minRate": 1.0, "maxRate": 5.0, "isUserSplitting": False, "isItemSplitting": False,
              "algoName": model, "foldInfo": "", "trainMatrix": train_ctx, "train": train2 if two_d else None,
              "rateDao": RateDao(prob["ui_user"], prob["ui_item"], prob["ctx_keys"]), "__enums__": ("Measure",)})
    # EmptyContextConditions: one ":na" condition per dimension, in header order (ContextRecommender.java:43) -- here the first of each
    conds_per_dim = nc // n_dims
    from oracle.jvm.interp import JCollection
    empty = [d * conds_per_dim for d in range(n_dims)]
    F.update({"EmptyContextConditions": JCollection([Box(e, "Integer") for e in empty]), "upbound": upbound, "lowbound": 1.0 / (10.0 ** 100),
              "isRankingPred": model.startswith("CAMF_") and model.endswith("CS"), "numF": NUM_F})
    if model == "SVD++":
        F["userItemsCache"] = UserItemsCache(cells2, nu)
    for name in STATE[model]:
        if name == "ccMatrix":
            sm = vm.new_object("librec/data/SymmMatrix")
            vm.call("librec/data/SymmMatrix", "<init>", "(I)V", [sm, nc])
            for a_ in range(nc):
                for b_ in range(a_, nc):
                    vm.call("librec/data/SymmMatrix", "set", "(IID)V", [sm, a_, b_, float(init[name][a_, b_])])
            F[JAVA_FIELD[name]] = sm
            continue
        if name in JAVA_FIELD:
            F[JAVA_FIELD[name]] = dense(vm, init[name]) if init[name].ndim == 2 else vector(vm, init[name])
            continue
        if model == "CAMF_CUCI":
            t = BoxTable()
            for r_ in range(init[name].shape[0]):
                for c_ in range(init[name].shape[1]):
                    t.put(r_, c_, init[name][r_, c_])
            F[name] = t
        else:
            F[name] = dense(vm, init[name]) if init[name].ndim == 2 else vector(vm, init[name])
    trace = []
    this.hooks["isConverged"] = lambda th, args: trace.append((th.fields["loss"], th.fields["lRate"]))
    if early_stop:   # `--early-stop RMSE|MAE`: isConverged() scores testMatrix after every epoch (IterativeRecommender.java:149-161)
        F["earlyStopMeasure"] = javasrc.EnumConst("Measure", early_stop)
        F["testMatrix"] = sparse(vm, len(prob["ui_user"]), len(prob["ctx_keys"]), test_cells)
        F["workingPath"] = ""
    if drop_in:
        F.update({"gpuHandle": javasrc.JLong(0), "fold": 1})
        javasrc.STATIC_FIELDS[("Recommender", "rateDao")] = F["rateDao"]
    this.call("buildModel", [])
    evals = None
    if test_cells:   # Recommender.evalRatings (Recommender.java:504-594) over a held-out testMatrix, from source as well
        F["testMatrix"] = sparse(vm, len(prob["ui_user"]), len(prob["ctx_keys"]), test_cells)
        F["workingPath"] = ""
        m = this.call("evalRatings", [])
        evals = {key.name: hx(val.v if isinstance(val, Box) else val) for key, val in m.d.items()}

    ranks = None
    if rank:         # Recommender.evalRankings (Recommender.java:672-955) from source, with carskit.eval.Measures from source over
        #              happy.coding.math.Measures / io.Lists / math.Stats from the happy.coding.utils jar's bytecode
        measures_cls = javasrc.This(vm, [os.path.join(ref, "src", "carskit", "eval", "Measures.java")], {})
        measures_cls.static_super = "happy/coding/math/Measures"
        this.class_map = dict(this.class_map, Measures=measures_cls, Lists="happy/coding/io/Lists", Stats="happy/coding/math/Stats")
        F["rateDao"] = source_dao(ref, vm, prob)
        javasrc.STATIC_FIELDS[("Recommender", "rateDao")] = F["rateDao"]
        F["testMatrix"] = sparse(vm, len(prob["ui_user"]), len(prob["ctx_keys"]), rank["test_cells"])
        F.update({"binThold": float(rank["bin_thold"]), "numRecs": int(rank["num_recs"]), "numIgnore": int(rank["num_ignore"]),
                  "isDiverseUsed": False, "evalStrategy": rank["strategy"], "workingPath": ""})
        before = this.statements
        m = this.call("evalRankings", [])
        ranks = {"measures": {key.name: hx(val.v if isinstance(val, Box) else val) for key, val in m.d.items()},
                 "statements": this.statements - before}

    def out_state(name):
        o = F[JAVA_FIELD.get(name, name)]
        if name == "ccMatrix":
            a = np.array([[vm.call("librec/data/SymmMatrix", "get", "(II)D", [o, a_, b_]) for b_ in range(nc)] for a_ in range(nc)])
            return [hx(x) for x in a.ravel()]
        if isinstance(o, BoxTable):
            a = np.array([[o.rows[Box(r_, "Integer")][Box(c_, "Integer")].v for c_ in range(init[name].shape[1])] for r_ in range(init[name].shape[0])])
        else:
            a = np.array(to_list(o.fields["data"]), dtype=np.float64)
        return [hx(x) for x in a.ravel()]
    rec = {"model": model, "k": k, "iters": iters, "bold_driver": bool(bold), "lrate": float(f32(lrate)), "regU": float(f32(reg)),
           "regI": float(f32(reg)), "regB": float(f32(reg)), "regC": float(f32(reg_c)), "global_mean": hx(gm),
           "empty_conds": empty, "n_ctx_dims": n_dims, "num_f": NUM_F,
           "problem": prob, "init": {n: [hx(x) for x in a.ravel()] for n, a in init.items()},
           "final": {n: out_state(n) for n in init}, "epoch_loss": [hx(l) for l, _ in trace], "epoch_lrate": [hx(r) for _, r in trace],
           "final_lrate": hx(F["lRate"]), "epochs_run": len(trace), "last_measure": hx(javasrc.unbox(F["last_measure"])), "test_cells": test_cells, "eval_ratings": evals, "rank": rank, "eval_rankings": ranks, "java_statements_executed": this.statements, "bytecode_instructions": vm.steps}
    return rec


def run_fm(ref, prob, k, iters, seed, reg_lw=0.01, reg_lf=0.02, rank=None):
    """FM.buildModel (FM.java:115-220: the dense ALS / coordinate-descent sweep) from source; state = (w0, w, V); rank: also
    Recommender.evalRankings() with FM.predict as the scorer"""
    vm = VM([os.path.join(ref, "lib", "librec-v1.4-alpha.jar"), os.path.join(ref, "lib", "happy.coding.utils-1.2.6.jar")] if rank else
            os.path.join(ref, "lib", "librec-v1.4-alpha.jar"))
    rng = np.random.default_rng(seed)
    nu, ni, nc = prob["n_users"], prob["n_items"], prob["n_conds"]
    p = nu + ni + nc
    size = len(prob["cells"])
    w, V = rng.random(p), 0.1 * rng.standard_normal((p, k))
    src = [os.path.join(ref, "src", "carskit", q) for q in ("alg/cars/adaptation/dependent/FM.java", "generic/ContextRecommender.java",
                                                              "generic/IterativeRecommender.java", "generic/Recommender.java")]
    this = javasrc.This(vm, src, CLASS_MAP)
    F = this.fields
    F.update({"w0": 0.0, "p": p, "k": k, "size": size, "w": vector(vm, w), "V": dense(vm, V), "Q": dense(vm, np.zeros((size, k))),
              "regLw": f32(reg_lw), "regLf": f32(reg_lf), "numFactors": k, "numIters": iters, "numUsers": nu, "numItems": ni,
              "numConditions": nc, "loss": 0.0, "trainMatrix": sparse(vm, len(prob["ui_user"]), len(prob["ctx_keys"]), prob["cells"]),
              "rateDao": RateDao(prob["ui_user"], prob["ui_item"], prob["ctx_keys"]), "verbose": False})
    this.call("buildModel", [])
    ranks = None
    if rank:
        measures_cls = javasrc.This(vm, [os.path.join(ref, "src", "carskit", "eval", "Measures.java")], {})
        measures_cls.static_super = "happy/coding/math/Measures"
        this.class_map = dict(this.class_map, Measures=measures_cls, Lists="happy/coding/io/Lists", Stats="happy/coding/math/Stats")
        F["rateDao"] = source_dao(ref, vm, prob)
        F.update({"testMatrix": sparse(vm, len(prob["ui_user"]), len(prob["ctx_keys"]), rank["test_cells"]), "binThold": float(rank["bin_thold"]),
                  "numRecs": int(rank["num_recs"]), "numIgnore": int(rank["num_ignore"]), "isDiverseUsed": False, "evalStrategy": rank["strategy"],
                  "workingPath": "", "isResultsOut": False, "isUserSplitting": False, "isItemSplitting": False, "algoName": "FM", "foldInfo": "",
                  "__enums__": ("Measure",)})
        m = this.call("evalRankings", [])
        ranks = {"measures": {key.name: hx(val.v if isinstance(val, Box) else val) for key, val in m.d.items()}}
    preds = []
    for ui, c, _ in prob["cells"][:12]:
        u_, j_ = prob["ui_user"][ui], prob["ui_item"][ui]
        preds.append([u_, j_, c, hx(this.call("predict", [u_, j_, c]))])
    return {"model": "FM", "k": k, "iters": iters, "regLw": float(f32(reg_lw)), "regLf": float(f32(reg_lf)), "problem": prob,
            "n_ctx_dims": len(prob["ctx_keys"][0].split(",")),
            "init": {"w": [hx(x) for x in w], "V": [hx(x) for x in V.ravel()]},
            "final": {"w0": hx(F["w0"]), "w": [hx(x) for x in to_list(F["w"].fields["data"])],
                      "V": [hx(x) for row in to_list(F["V"].fields["data"]) for x in row]},
            "final_loss": hx(F["loss"]), "predictions": preds, "rank": rank, "eval_rankings": ranks,
            "java_statements_executed": this.statements, "bytecode_instructions": vm.steps}


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    rng = np.random.default_rng(20260928)
    out = {"source": "reference Java SOURCE (src/carskit/**) interpreted by oracle/jvm/javasrc.py over the librec jar's bytecode "
                     "(oracle/jvm/interp.py); doubles are C99 hex strings", "cases": []}
    for model in MODELS:
        for (nu, ni, nd, cpd, n, k, iters) in ((7, 5, 2, 3, 60, 3, 4), (12, 9, 3, 2, 150, 10, 3)):
            prob = problem(rng, nu, ni, nd, cpd, n + n // 4)
            held = [c for i, c in enumerate(prob["cells"]) if i % 5 == 4]           # every fifth cell is test data
            prob["cells"] = [c for i, c in enumerate(prob["cells"]) if i % 5 != 4]
            rec = run_model(ref, model, prob, k, iters, seed=int(rng.integers(1 << 30)), test_cells=held)
            out["cases"].append(rec)
            print("%-10s k=%-3d %d ratings, %d epochs: loss %s -> %s   (%d Java statements, %d bytecode instructions)"
                  % (model, k, len(prob["cells"]), iters, float.fromhex(rec["epoch_loss"][0]), float.fromhex(rec["epoch_loss"][-1]),
                     rec["java_statements_executed"], rec["bytecode_instructions"]), flush=True)
    out["fm_cases"] = []
    for (nu, ni, nd, cpd, n, k, iters) in ((4, 3, 2, 2, 14, 2, 2), (5, 4, 2, 3, 24, 3, 2)):
        prob = problem(rng, nu, ni, nd, cpd, n)
        rec = run_fm(ref, prob, k, iters, seed=int(rng.integers(1 << 30)))
        out["fm_cases"].append(rec)
        print("FM         k=%-3d %d ratings, %d sweeps   (%d Java statements, %d bytecode instructions)"
              % (k, len(prob["cells"]), iters, rec["java_statements_executed"], rec["bytecode_instructions"]), flush=True)
    path = os.path.join(ROOT, "tests", "golden", "reference_src.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path)


if __name__ == "__main__":
    main()

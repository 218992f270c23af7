#!/usr/bin/env python3
"""Execute THIS repository's source-only Java drop-in classes (java/carskit/alg/gpu/*_GPU.java, GpuSupport, Dev, Rows) -- build container only.

    python oracle/check_java_binding.py [/root/reference]

There is no JDK in the image, so the Java side of the boundary was only ever checked as text.  Here it RUNS: the drop-in
class is put in front of the reference's own class chain (X_GPU extends X extends ... Recommender, the reference's files read where they
lie) and interpreted by oracle/jvm/javasrc.py; `buildModel()` therefore goes through `GpuSupport.buildModel(this)` -- the marshalling
(`pairMaps`, `contextTable`, `Rows.of`, `Dev.set*/get*`), the epoch loop with the reference's UNCHANGED `isConverged()` (bold driver), the
copy-back -- and every `NativeMF.*` native lands in a stand-in for the JNI shim that drives the order-exact CPU oracle with exactly the
arguments the C ABI would get (the real natives need a GPU; the arguments are what is under test).

Bar: for the same problems as tests/golden/reference_src.json the containers the drop-in leaves in the Java objects, every epoch's loss and
every bold-driver rate are BIT-IDENTICAL to what the reference's own buildModel() produced -- a wrong container id, a swapped
regulariser, a stale copy-in or a missing copy-back cannot pass.  Writes tests/golden/java_binding_check.json (the native call sequence
per model + the verdict); tests/test_java_binding_exec.py re-runs it when /root/reference is present."""
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mint_reference_src as M  # noqa: E402
from oracle import oracle_c  # noqa: E402
from oracle.jvm import javasrc  # noqa: E402
from oracle.jvm.interp import JArray  # noqa: E402

JAVA = os.path.join(ROOT, "java", "carskit", "alg", "gpu")
STATE_NAMES = {0: "P", 1: "Q", 2: "userBias", 3: "itemBias", 4: "condBias", 5: "ucBias", 6: "icBias", 7: "Y", 8: "ccMatrix", 9: "cfMatrix",
               10: "cVector"}
MODEL_NAMES = {0: "BiasedMF", 1: "CAMF_C", 2: "CAMF_CI", 3: "CAMF_CU", 4: "CAMF_CUCI", 5: "PMF", 6: "SVD++", 7: "CAMF_ICS", 8: "CAMF_LCS",
               9: "CAMF_MCS"}
DROP_IN = {"BiasedMF": "BiasedMF_GPU", "PMF": "PMF_GPU", "CAMF_C": "CAMF_C_GPU", "CAMF_CI": "CAMF_CI_GPU", "CAMF_CU": "CAMF_CU_GPU",
           "CAMF_CUCI": "CAMF_CUCI_GPU", "SVD++": "SVDPP_GPU", "CAMF_ICS": "CAMF_ICS_GPU", "CAMF_LCS": "CAMF_LCS_GPU", "CAMF_MCS": "CAMF_MCS_GPU"}


def arr(a, dtype):
    return np.array(a.data if isinstance(a, JArray) else a, dtype=dtype)


def native_constants():
    """`public static final int A = 1, B = 0x80;` of NativeMF.java"""
    text = open(os.path.join(JAVA, "NativeMF.java")).read()
    out = {}
    for decl in re.findall(r"public static final int ([^;]+);", text):
        for part in decl.split(","):
            name, val = part.split("=")
            out[name.strip()] = int(val.strip(), 0)
    return out


class Natives:
    """stand-in for jni/carskit_jni.cpp + the library: one order-exact CPU oracle per handle"""

    def __init__(self):
        self.consts = native_constants()
        self.h, self.calls, self.live = {}, [], 0

    def jstatic(self, name, args):
        self.calls.append(name)
        return getattr(self, "n_" + name)(*[javasrc.unbox(a) for a in args])

    def n_create(self, model, k, nu, ni, nc, device, flags):
        self.live += 1
        hid = 1000 + len(self.h)
        self.h[hid] = {"model": MODEL_NAMES[int(model)], "k": int(k), "nu": int(nu), "ni": int(ni), "nc": int(nc), "flags": int(flags),
                       "state": {}, "orc": None, "sim": None}
        assert device == 0
        return javasrc.JLong(hid)

    def n_destroy(self, h):
        self.live -= 1
        self.h[int(h)]["destroyed"] = True

    def n_setSimParams(self, h, num_f, n_dims, empty):
        self.h[int(h)]["sim"] = (int(num_f), int(n_dims), arr(empty, np.int32))

    def n_setRatingsCsr(self, h, row_ptr, col_ind, data, ui_user, ui_item, ctx_ptr, ctx_conds):
        d = self.h[int(h)]
        rp, ci, v, uu, ui = arr(row_ptr, np.int64), arr(col_ind, np.int32), arr(data, np.float64), arr(ui_user, np.int32), arr(ui_item, np.int32)
        rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
        d["tuples"] = (uu[rows], ui[rows], ci, v, arr(ctx_ptr, np.int32), arr(ctx_conds, np.int32))

    def n_setRatings2D(self, h, row_ptr, col_ind, data):
        d = self.h[int(h)]
        rp, ci, v = arr(row_ptr, np.int64), arr(col_ind, np.int32), arr(data, np.float64)
        rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp)).astype(np.int32)
        d["tuples"] = (rows, ci, np.zeros(len(ci), np.int32), v, np.zeros(1, np.int32), np.zeros(0, np.int32))

    def n_setHparams(self, h, reg_u, reg_i, reg_b, reg_c, gm):
        self.h[int(h)]["hp"] = (float(reg_u), float(reg_i), float(reg_b), float(reg_c), float(gm))

    def n_setMatrix(self, h, which, rows):
        self.h[int(h)]["state"][STATE_NAMES[int(which)]] = np.array([arr(r, np.float64) for r in rows])

    def n_setVector(self, h, which, v):
        self.h[int(h)]["state"][STATE_NAMES[int(which)]] = arr(v, np.float64)

    def _oracle(self, d):
        if d["orc"] is None:
            u, j, ctx, r, ctx_ptr, ctx_conds = d["tuples"]
            ru, ri, rb, rc, gm = d["hp"]
            if d["model"] in ("SVD++", "CAMF_ICS", "CAMF_LCS", "CAMF_MCS"):
                num_f, n_dims, empty = d["sim"] if d["sim"] else (0, 1, np.zeros(0, np.int32))     # SVD++: no context dimensions in play
                d["orc"] = oracle_c.SimOracle(d["model"], d["k"], d["nu"], d["ni"], d["nc"], u, j, ctx, r, ctx_ptr, ctx_conds, empty, d["state"], gm,
                                              ru, ri, rb, rc, n_ctx_dims=n_dims)
            else:
                d["orc"] = oracle_c.Oracle(d["model"], d["k"], d["nu"], d["ni"], d["nc"], u, j, ctx, r, ctx_ptr, ctx_conds, d["state"], gm, ru, ri,
                                           rb, rc)
        return d["orc"]

    def n_setEvalRatings(self, h, u, j, ctx, r):
        self.h[int(h)]["eval"] = (arr(u, np.int32), arr(j, np.int32), None if ctx is None else arr(ctx, np.int32), arr(r, np.float64))

    def n_evalResident(self, h, min_rate, max_rate):
        d = self.h[int(h)]
        u, j, ctx, r = d["eval"]
        ev = self._oracle(d).eval_ratings(u, j, ctx if ctx is not None else np.zeros(len(u), np.int32), r, float(min_rate), float(max_rate))
        return JArray("D", [float(ev[n]) for n in ("MAE", "RMSE", "NMAE", "rMAE", "rRMSE")])

    def n_evalRankings(self, h, tu, tj, tctx, tr, su, sj, sctx, sr, bin_thold, num_recs, num_ignore, strategy):
        """cmi_eval_rankings: (train tuples, test tuples, threshold, topN, numIgnore, strategy) -> the 21 measures in the order
        GpuSupport.rankingMeasures unpacks them"""
        from oracle import rank_oracle
        d = self.h[int(h)]
        orc = self._oracle(d)
        tup = lambda u, j, c, r: list(zip(arr(u, np.int32).tolist(), arr(j, np.int32).tolist(), arr(c, np.int32).tolist(), arr(r, np.float64).tolist()))
        got, _ = rank_oracle.eval_rankings(lambda a_, b_, c_: orc.predict(a_, b_, c_), tup(tu, tj, tctx, tr), tup(su, sj, sctx, sr),
                                           bin_thold=float(bin_thold), num_recs=int(num_recs), num_ignore=int(num_ignore),
                                           strategy="uc" if int(strategy) == self.consts["RANK_UC"] else "ucu")
        order = ("Pre5", "Pre10", "PreN", "Rec5", "Rec10", "RecN", "AUC5", "AUC10", "AUCN", "MAP5", "MAP10", "MAPN", "NDCG5", "NDCG10", "NDCGN",
                 "MRR5", "MRR10", "MRRN", "D5", "D10", "DN")
        return JArray("D", [float(got[m]) for m in order])

    def n_trainEpoch(self, h, lrate):
        return float(self._oracle(self.h[int(h)]).epoch(float(lrate)))

    def _out(self, h, which):
        d = self.h[int(h)]
        return np.asarray((d["orc"].state if d["orc"] is not None else d["state"])[STATE_NAMES[int(which)]], dtype=np.float64)

    def n_getMatrix(self, h, which, rows):
        a = self._out(h, which)
        a = a.reshape(len(rows), -1)
        for row, src in zip(rows, a):
            tgt = row.data if isinstance(row, JArray) else row
            assert len(tgt) == len(src)
            tgt[:] = [float(x) for x in src]

    def n_getVector(self, h, which, v):
        tgt = v.data if isinstance(v, JArray) else v
        src = self._out(h, which).ravel()
        assert len(tgt) == len(src)
        tgt[:] = [float(x) for x in src]


class FMNatives(Natives):
    """the fm* natives over oracle/carskit_oracle_fm.c"""

    def n_fmCreate(self, k, nu, ni, nc, n_dims, device, flags):
        self.live += 1
        hid = 2000 + len(self.h)
        self.h[hid] = {"k": int(k), "nu": int(nu), "ni": int(ni), "nc": int(nc), "dims": int(n_dims)}
        return javasrc.JLong(hid)

    def n_fmDestroy(self, h):
        self.live -= 1

    def n_fmSetHparams(self, h, reg_lw, reg_lf, global_size):
        self.h[int(h)]["hp"] = (float(reg_lw), float(reg_lf), int(global_size))

    def n_fmSetRatingsCsr(self, h, row_ptr, col_ind, data, ui_user, ui_item):
        rp, ci, v, uu, ui = arr(row_ptr, np.int64), arr(col_ind, np.int32), arr(data, np.float64), arr(ui_user, np.int32), arr(ui_item, np.int32)
        rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
        self.h[int(h)]["tuples"] = (uu[rows], ui[rows], ci, v)

    def n_fmSetModel(self, h, w0, w, v_rows):
        self.h[int(h)]["model"] = (float(w0), arr(w, np.float64), np.array([arr(r, np.float64) for r in v_rows]))

    def n_fmTrain(self, h, iters):
        d = self.h[int(h)]
        u, j, ctx, r = d["tuples"]
        assert d["hp"][2] == len(r)
        w0, w, V = d["model"]
        orc = oracle_c.FMOracle(d["k"], d["nu"], d["ni"], d["nc"], d["dims"], u, j, ctx, r, w0, w, V, d["hp"][0], d["hp"][1])
        orc.init()
        for _ in range(int(iters)):
            orc.sweep()
        d["orc"] = orc

    def n_fmEvalRankings(self, h, tu, tj, tctx, tr, su, sj, sctx, sr, bin_thold, num_recs, num_ignore, strategy):
        from oracle import rank_oracle
        d = self.h[int(h)]
        u, j, ctx, r = d["tuples"]
        w0, w, V = d["model"]
        orc = oracle_c.FMOracle(d["k"], d["nu"], d["ni"], d["nc"], d["dims"], u, j, ctx, r, w0, w, V, d["hp"][0], d["hp"][1])
        tup = lambda a_, b_, c_, r_: list(zip(arr(a_, np.int32).tolist(), arr(b_, np.int32).tolist(), arr(c_, np.int32).tolist(), arr(r_, np.float64).tolist()))
        got, _ = rank_oracle.eval_rankings(lambda a_, b_, c_: orc.predict(a_, b_, c_), tup(tu, tj, tctx, tr), tup(su, sj, sctx, sr),
                                           bin_thold=float(bin_thold), num_recs=int(num_recs), num_ignore=int(num_ignore),
                                           strategy="uc" if int(strategy) == self.consts["RANK_UC"] else "ucu")
        order = ("Pre5", "Pre10", "PreN", "Rec5", "Rec10", "RecN", "AUC5", "AUC10", "AUCN", "MAP5", "MAP10", "MAPN", "NDCG5", "NDCG10", "NDCGN",
                 "MRR5", "MRR10", "MRRN", "D5", "D10", "DN")
        return JArray("D", [float(got[m]) for m in order])

    def n_fmGetModel(self, h, w, v_rows):
        orc = self.h[int(h)]["orc"]
        (w.data if isinstance(w, JArray) else w)[:] = [float(x) for x in orc.w]
        for row, src in zip(v_rows, orc.V):
            (row.data if isinstance(row, JArray) else row)[:] = [float(x) for x in src]
        return float(orc.w0)


def check_fm_rank(ref, case):
    """FM_GPU.evalRankings() with -Dcarskit.gpu.rank=true on the reference's trained FM model, against the reference's own evalRankings()"""
    import math
    from oracle.jvm.interp import VM
    vm = VM([os.path.join(ref, "lib", "librec-v1.4-alpha.jar"), os.path.join(ref, "lib", "happy.coding.utils-1.2.6.jar")])
    prob, k, rk = case["problem"], case["k"], case["rank"]
    nu, ni, nc = prob["n_users"], prob["n_items"], prob["n_conds"]
    nat = FMNatives()
    cmap = dict(M.CLASS_MAP, NativeMF=nat)
    for n in ("GpuSupport", "Dev", "Rows"):
        cmap[n] = static_class(vm, n, cmap)
    src = [os.path.join(JAVA, "FM_GPU.java")] + [os.path.join(ref, "src", "carskit", "generic", q) for q in
                                                   ("ContextRecommender.java", "IterativeRecommender.java", "Recommender.java")]
    this = javasrc.This(vm, src, cmap)
    fin = case["final"]
    V = np.array([float.fromhex(x) for x in fin["V"]]).reshape(nu + ni + nc, k)
    dao = M.source_dao(ref, vm, prob)
    javasrc.STATIC_FIELDS[("Recommender", "rateDao")] = dao
    this.fields.update({"w0": float.fromhex(fin["w0"]), "p": nu + ni + nc, "k": k, "w": M.vector(vm, [float.fromhex(x) for x in fin["w"]]),
                        "V": M.dense(vm, V), "regLw": M.f32(case["regLw"]), "regLf": M.f32(case["regLf"]), "numFactors": k, "numUsers": nu,
                        "numItems": ni, "numConditions": nc, "fold": 1, "rateDao": dao, "isDiverseUsed": False,
                        "trainMatrix": M.sparse(vm, len(prob["ui_user"]), len(prob["ctx_keys"]), prob["cells"]),
                        "testMatrix": M.sparse(vm, len(prob["ui_user"]), len(prob["ctx_keys"]), rk["test_cells"]),
                        "binThold": float(rk["bin_thold"]), "numRecs": int(rk["num_recs"]), "numIgnore": int(rk["num_ignore"]),
                        "evalStrategy": rk["strategy"], "__enums__": ("Measure",)})
    javasrc.SYSTEM_PROPERTIES["carskit.gpu.rank"] = "true"
    try:
        m = this.call("evalRankings", [])
    finally:
        javasrc.SYSTEM_PROPERTIES.pop("carskit.gpu.rank", None)
    assert nat.live == 0
    same = {}
    for key, val in m.d.items():
        a, b = float(javasrc.unbox(val)), float.fromhex(case["eval_rankings"]["measures"][key.name])
        same[key.name] = (a == b) or (math.isnan(a) and math.isnan(b)) or (key.name.startswith("NDCG") and abs(a - b) <= 4 * math.ulp(b))
    return nat.calls, same


def check_fm(ref, case):
    """FM_GPU.buildModel() + predict() against the reference's FM.buildModel() / predict() (tests/golden/reference_src.json, fm_cases)"""
    from oracle.jvm.interp import VM, to_list
    vm = VM(os.path.join(ref, "lib", "librec-v1.4-alpha.jar"))
    prob, k = case["problem"], case["k"]
    nu, ni, nc = prob["n_users"], prob["n_items"], prob["n_conds"]
    nat = FMNatives()
    cmap = dict(M.CLASS_MAP, NativeMF=nat)
    for n in ("GpuSupport", "Dev", "Rows"):
        cmap[n] = static_class(vm, n, cmap)
    src = [os.path.join(JAVA, "FM_GPU.java")] + [os.path.join(ref, "src", "carskit", "generic", q) for q in
                                                   ("ContextRecommender.java", "IterativeRecommender.java", "Recommender.java")]
    this = javasrc.This(vm, src, cmap)
    w = [float.fromhex(x) for x in case["init"]["w"]]
    V = np.array([float.fromhex(x) for x in case["init"]["V"]]).reshape(nu + ni + nc, k)
    this.fields.update({"w0": 0.0, "p": nu + ni + nc, "k": k, "w": M.vector(vm, w), "V": M.dense(vm, V), "regLw": M.f32(case["regLw"]),
                        "regLf": M.f32(case["regLf"]), "numFactors": k, "numIters": case["iters"], "numUsers": nu, "numItems": ni,
                        "numConditions": nc, "fold": 1, "trainMatrix": M.sparse(vm, len(prob["ui_user"]), len(prob["ctx_keys"]), prob["cells"]),
                        "rateDao": M.RateDao(prob["ui_user"], prob["ui_item"], prob["ctx_keys"])})
    this.call("buildModel", [])
    F = this.fields
    same = {"w0": float(F["w0"]).hex() == case["final"]["w0"],
            "w": [float(x).hex() for x in to_list(F["w"].fields["data"])] == case["final"]["w"],
            "V": [float(x).hex() for row in to_list(F["V"].fields["data"]) for x in row] == case["final"]["V"]}
    worst = 0.0
    for u_, j_, c_, want in case["predictions"]:       # the pairwise form of the model equation: same value to rounding, not to the bit
        worst = max(worst, abs(this.call("predict", [u_, j_, c_]) - float.fromhex(want)))
    same["predict_within_1e-12"] = worst <= 1e-12
    assert nat.live == 0
    return nat.calls, same, this.statements


def static_class(vm, name, class_map):
    t = javasrc.This(vm, [os.path.join(JAVA, name + ".java")], class_map)
    return t


def check(ref, case):
    """run the drop-in for one golden case; returns (native call names, verdict dict)"""
    model = case["model"]
    nat = Natives()
    cmap = {"NativeMF": nat}
    holder = {}

    def make(vm):
        for n in ("GpuSupport", "Dev", "Rows"):
            holder[n] = static_class(vm, n, cmap)
            cmap[n] = holder[n]
    init = {n: [float.fromhex(x) for x in v] for n, v in case["init"].items()}
    rec = M.run_model(ref, model, case["problem"], case["k"], case["iters"], seed=0, lrate=case["lrate"], bold=case["bold_driver"],
                      drop_in=(os.path.join(JAVA, DROP_IN[model] + ".java"), cmap, make), init_override=init)
    same = {n: rec["final"][n] == case["final"][n] for n in case["final"]}
    same["epoch_loss"] = rec["epoch_loss"] == case["epoch_loss"]
    same["epoch_lrate"] = rec["epoch_lrate"] == case["epoch_lrate"]
    assert nat.live == 0, "the native handle was not destroyed"
    return nat.calls, same, rec["java_statements_executed"]


class GroupNatives(Natives):
    """the group* natives (cmi_group_*): W user shards, each an order-exact CPU oracle; after every epoch the item-side containers become
    start + mean of the shards' moves, the local rate is lRate x the scale set by groupSetLrScale"""
    USER_SIDE = ("P", "userBias", "ucBias")

    def n_groupCreate(self, model, k, nu, ni, nc, n_shards, devices, flags):
        self.live += 1
        gid = 5000 + len(self.h)
        self.h[gid] = {"model": MODEL_NAMES[int(model)], "k": int(k), "nu": int(nu), "ni": int(ni), "nc": int(nc), "W": int(n_shards),
                       "state": {}, "scale": 1.0, "shards": None}
        assert devices is None
        return javasrc.JLong(gid)

    def n_groupDestroy(self, g):
        self.live -= 1

    def n_groupSetHparams(self, g, ru, ri, rb, rc, gm):
        self.n_setHparams(g, ru, ri, rb, rc, gm)

    def n_groupSetRatingsCsr(self, g, *a):
        self.n_setRatingsCsr(g, *a)

    def n_groupSetRatings2D(self, g, *a):
        self.n_setRatings2D(g, *a)

    def n_groupSetMatrix(self, g, which, rows):
        self.n_setMatrix(g, which, rows)

    def n_groupSetVector(self, g, which, v):
        self.n_setVector(g, which, v)

    def n_groupSetLrScale(self, g, scale):
        self.h[int(g)]["scale"] = float(scale)

    def _shards(self, d):
        if d["shards"] is None:
            from carskit_amd import dist as cdist
            from carskit_amd.synth import RatingData
            u, j, ctx, r, ctx_ptr, ctx_conds = d["tuples"]
            data = RatingData(d["nu"], d["ni"], d["nc"], 1, u.astype(np.int32), j.astype(np.int32), ctx.astype(np.int32), r, ctx_ptr, ctx_conds)
            ru, ri, rb, rc, gm = d["hp"]
            d["shards"] = []
            for rank in range(d["W"]):
                sh, (lo, hi) = cdist.shard_by_user(data, rank, d["W"])
                st = {n: (a[lo:hi].copy() if n in self.USER_SIDE else a.copy()) for n, a in d["state"].items()}
                orc = oracle_c.Oracle(d["model"], d["k"], hi - lo, d["ni"], d["nc"], sh.u, sh.j, sh.ctx, sh.r, ctx_ptr, ctx_conds, st, gm, ru, ri,
                                      rb, rc)
                d["shards"].append((orc, lo, hi))
        return d["shards"]

    def n_groupTrainEpoch(self, g, lrate):
        d = self.h[int(g)]
        shards = self._shards(d)
        names = [n for n in d["state"] if n not in self.USER_SIDE]
        start = {n: shards[0][0].state[n].copy() for n in names}
        loss = 0.0
        for orc, _, _ in shards:
            loss += float(orc.epoch(float(lrate) * d["scale"]))
        for n in names:
            merged = start[n] + sum(o.state[n] - start[n] for o, _, _ in shards) / float(d["W"])
            for o, _, _ in shards:
                o.state[n][...] = merged
        return loss

    def _gather(self, g, which):
        d = self.h[int(g)]
        name = STATE_NAMES[int(which)]
        if d["shards"] is None:
            return d["state"][name]
        if name in self.USER_SIDE:
            return np.concatenate([np.asarray(o.state[name]).reshape(hi - lo, -1) for o, lo, hi in d["shards"]]).reshape(d["state"][name].shape)
        return d["shards"][0][0].state[name]

    def n_groupGetMatrix(self, g, which, rows):
        a = np.asarray(self._gather(g, which), dtype=np.float64).reshape(len(rows), -1)
        for row, src in zip(rows, a):
            (row.data if isinstance(row, JArray) else row)[:] = [float(x) for x in src]

    def n_groupGetVector(self, g, which, v):
        (v.data if isinstance(v, JArray) else v)[:] = [float(x) for x in np.asarray(self._gather(g, which)).ravel()]


def check_group(ref, case, n_shards=2):
    """-Dcarskit.shards=N: GpuSupport.buildModelSharded -- groupCreate / groupSet* / Dev.ofGroup routing of copyIn and copyOut (negative
    handles) / groupSetLrScale(sqrt(N)) / the epoch loop over groupTrainEpoch with the reference's isConverged() steering the base rate.
    Expected: the same natives driven by a few lines of Python in the intended order, with IterativeRecommender's bold driver restated."""
    model = case["model"]
    init = {n: np.array([float.fromhex(x) for x in v]) for n, v in case["init"].items()}
    nat = GroupNatives()
    cmap = {"NativeMF": nat}

    def make(vm):
        for n in ("GpuSupport", "Dev", "Rows"):
            cmap[n] = static_class(vm, n, cmap)
    javasrc.SYSTEM_PROPERTIES["carskit.shards"] = str(n_shards)
    try:
        got = M.run_model(ref, model, case["problem"], case["k"], case["iters"], seed=0, lrate=case["lrate"], bold=case["bold_driver"],
                          drop_in=(os.path.join(JAVA, DROP_IN[model] + ".java"), cmap, make),
                          init_override={n: a.tolist() for n, a in init.items()})
    finally:
        javasrc.SYSTEM_PROPERTIES.pop("carskit.shards", None)
    assert nat.live == 0
    # the intended flow, in Python, over a second set of the same stand-ins
    ref_nat = GroupNatives()
    first = [h for h in nat.h.values()][0]
    p = case["problem"]
    shapes = {"P": (p["n_users"], case["k"]), "Q": (p["n_items"], case["k"]), "userBias": (p["n_users"],), "itemBias": (p["n_items"],),
              "ucBias": (p["n_users"], p["n_conds"]), "icBias": (p["n_items"], p["n_conds"])}
    g = ref_nat.n_groupCreate({v: k_ for k_, v in MODEL_NAMES.items()}[model], case["k"], p["n_users"], p["n_items"], p["n_conds"], n_shards, None, 0)
    ref_nat.h[int(g)]["hp"] = first["hp"]
    ref_nat.h[int(g)]["tuples"] = first["tuples"]
    ref_nat.h[int(g)]["state"] = {n: init[n].reshape(shapes[n]).copy() for n in init}
    ref_nat.n_groupSetLrScale(g, float(np.sqrt(float(n_shards))))
    lr, last, losses, lrates = case["lrate"], 0.0, [], []
    for it in range(1, case["iters"] + 1):
        lrates.append(lr)
        loss = ref_nat.n_groupTrainEpoch(g, lr)
        losses.append(loss)
        if it > 1:                                   # IterativeRecommender.updateLRate, bold driver (IterativeRecommender.java:216-229)
            lr = lr * 1.05 if abs(last) > abs(loss) else lr * 0.5
        last = loss
    same = {"epoch_loss": got["epoch_loss"] == [M.hx(x) for x in losses], "epoch_lrate": got["epoch_lrate"] == [M.hx(x) for x in lrates]}
    which = {v: k_ for k_, v in STATE_NAMES.items()}
    for n in case["final"]:
        want = np.asarray(ref_nat._gather(g, which[n]), dtype=np.float64).ravel()
        same[n] = got["final"][n] == [M.hx(x) for x in want]
    return nat.calls, same


def check_rank(ref, case):
    """-Dcarskit.gpu.rank=true: the drop-in's evalRankings() -> GpuSupport.evalRankings (fresh handle, upload, tuples of the train and test
    matrices, NativeMF.evalRankings, rankingMeasures) against the reference's own evalRankings() of the same model
    (tests/golden/reference_rank.json); nDCG within 4 ulp (Math.log), everything else bit for bit."""
    import math
    model = case["model"]
    init = {n: [float.fromhex(x) for x in v] for n, v in case["init"].items()}
    nat = Natives()
    cmap = {"NativeMF": nat}

    def make(vm):
        for n in ("GpuSupport", "Dev", "Rows"):
            cmap[n] = static_class(vm, n, cmap)
    javasrc.SYSTEM_PROPERTIES["carskit.gpu.rank"] = "true"
    try:
        got = M.run_model(ref, model, case["problem"], case["k"], case["iters"], seed=0, lrate=case["lrate"], bold=case["bold_driver"],
                          drop_in=(os.path.join(JAVA, DROP_IN[model] + ".java"), cmap, make), init_override=init, rank=case["rank"])
    finally:
        javasrc.SYSTEM_PROPERTIES.pop("carskit.gpu.rank", None)
    assert nat.live == 0
    same = {}
    for m, want in case["eval_rankings"]["measures"].items():
        a, b = float.fromhex(got["eval_rankings"]["measures"][m]), float.fromhex(want)
        same[m] = (a == b) or (math.isnan(a) and math.isnan(b)) or (m.startswith("NDCG") and abs(a - b) <= 4 * math.ulp(b))
    return nat.calls, same


def check_early_stop(ref, case, measure="RMSE"):
    """`--early-stop RMSE`: the reference's isConverged() calls evalRatings() after every epoch; the drop-in's override answers from the
    live native model (GpuSupport.evalResident) -- setEvalRatings / tuples() / evaluatesDuringTraining() / the handle bookkeeping.  The
    reference's own run with the same setting (interpreted, no drop-in) must stop at the same epoch with the same model."""
    model = case["model"]
    init = {n: [float.fromhex(x) for x in v] for n, v in case["init"].items()}
    kw = dict(seed=0, lrate=case["lrate"], bold=case["bold_driver"], init_override=init, test_cells=case["test_cells"], early_stop=measure)
    want = M.run_model(ref, model, case["problem"], case["k"], case["iters"], **kw)
    want.pop("eval_ratings", None)
    nat = Natives()
    cmap = {"NativeMF": nat}

    def make(vm):
        for n in ("GpuSupport", "Dev", "Rows"):
            cmap[n] = static_class(vm, n, cmap)
    got = M.run_model(ref, model, case["problem"], case["k"], case["iters"], drop_in=(os.path.join(JAVA, DROP_IN[model] + ".java"), cmap, make), **kw)
    same = {n: got["final"][n] == want["final"][n] for n in want["final"]}
    for key in ("epoch_loss", "epoch_lrate", "epochs_run", "last_measure"):
        same[key] = got[key] == want[key]
    assert nat.live == 0
    return nat.calls, same


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_src.json")))["cases"]
    out = {"note": "oracle/check_java_binding.py: the source-only Java drop-ins executed by oracle/jvm/javasrc.py over the reference's class "
                   "chain, natives -> CPU oracle; bit-identical to the reference's own buildModel() (tests/golden/reference_src.json)",
           "models": {}}
    ok = True
    for case in cases:
        if case["k"] != min(c["k"] for c in cases if c["model"] == case["model"]):
            continue
        calls, same, stmts = check(ref, case)
        out["models"][case["model"]] = {"drop_in": DROP_IN[case["model"]], "native_calls": calls, "bit_identical": same,
                                        "java_statements_executed": stmts}
        ok = ok and all(same.values())
        print("%-10s %-14s %s  (%d natives, %d statements)" % (case["model"], DROP_IN[case["model"]], "bit-identical" if all(same.values())
              else "DIFFERS: %s" % [n for n, v in same.items() if not v], len(calls), stmts), flush=True)
    es_case = [c for c in cases if c["model"] == "CAMF_CU"][0]
    calls, same = check_early_stop(ref, es_case)
    out["early_stop_rmse"] = {"model": "CAMF_CU", "native_calls": calls, "bit_identical": same}
    ok = ok and all(same.values())
    print("early stop on RMSE (CAMF_CU_GPU):", "bit-identical" if all(same.values()) else "DIFFERS %s" % same, "natives:", sorted(set(calls)), flush=True)
    rank_cases = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_rank.json")))["cases"]
    out["rank_on_gpu"] = {}
    for rc in rank_cases:
        calls, same = check_rank(ref, rc)
        out["rank_on_gpu"][rc["model"]] = {"native_calls": calls, "same_measures": same}
        ok = ok and all(same.values())
        print("-Dcarskit.gpu.rank (%s):" % DROP_IN[rc["model"]], "the reference's measures" if all(same.values()) else "DIFFERS %s" % [m for m, v in same.items() if not v], flush=True)
    fm_rank_case = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_rank.json")))["fm_cases"][0]
    calls, same = check_fm_rank(ref, fm_rank_case)
    out["rank_on_gpu"]["FM"] = {"native_calls": calls, "same_measures": same}
    ok = ok and all(same.values())
    print("-Dcarskit.gpu.rank (FM_GPU):", "the reference's measures" if all(same.values()) else "DIFFERS %s" % [m for m, v in same.items() if not v], flush=True)
    gr_case = [c for c in cases if c["model"] == "CAMF_CI"][0]
    calls, same = check_group(ref, gr_case)
    out["shards_2"] = {"model": "CAMF_CI", "native_calls": calls, "bit_identical": same}
    ok = ok and all(same.values())
    print("-Dcarskit.shards=2 (CAMF_CI_GPU):", "as intended, bit for bit" if all(same.values()) else "DIFFERS %s" % same, "natives:", sorted(set(calls)), flush=True)
    fm_case = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_src.json")))["fm_cases"][0]
    calls, same, stmts = check_fm(ref, fm_case)
    out["models"]["FM"] = {"drop_in": "FM_GPU", "native_calls": calls, "bit_identical": same, "java_statements_executed": stmts}
    ok = ok and all(same.values())
    print("%-10s %-14s %s  (%d natives, %d statements)" % ("FM", "FM_GPU", "bit-identical model, predict within 1e-12" if all(same.values())
          else "DIFFERS: %s" % [n for n, v in same.items() if not v], len(calls), stmts), flush=True)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "java_binding_check.json"), "w"), indent=0)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

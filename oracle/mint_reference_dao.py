#!/usr/bin/env python3
"""Mint tests/golden/reference_dao.json by EXECUTING the reference's own `DataDAO.readData` (build container only).

    python oracle/mint_reference_dao.py [/root/reference]

SURVEY 8(f) N2 / A10: the id-mapper either side of the hot path.  `readData(double)` (src/carskit/data/processor/DataDAO.java:166-412) and
the `num*()` getters are run statement by statement from the text of DataDAO.java by oracle/jvm/javasrc.py; the `new SparseMatrix(rows,
cols, dataTable, colMap)` at its end goes into the vendored librec jar's BYTECODE (carskit.data.structure.SparseMatrix only forwards to
that constructor), which fixes the CRS order of the rating cells.  What this script provides is what the constructor (DataDAO.java:119-149)
would: empty maps for the fields, or -- for a test file read after its training file, the reference's `-testset` flow -- the maps of the
DAO read before it.

Inputs: the two binary sample files under tests/golden/ (the reference's own sampleData) and a few files this script generates (clean and
"messy": tabs in the header, padded keys, ratings like `4.0d`, rows with no active condition, CRLF).  Outputs are data: raw ids in inner-id
order, the (ui, ctx, rating) cells in CRS order, the condition lists, EmptyContextConditions, ratingScale.  tests/test_reference_dao.py
requires the product's C++ DataDAO (through the C ABI) and oracle/dao_oracle.py to reproduce them exactly."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.jvm import javasrc  # noqa: E402
from oracle.jvm.interp import VM, Box, GuavaMultimap, JCollection, JString  # noqa: E402

CLASS_MAP = {"SparseMatrix": "librec/data/SparseMatrix"}
SM = "librec/data/SparseMatrix"
SHARED = ("userIds", "itemIds", "ctxIds", "uiIds", "dimIds", "condIds", "uRatedList", "iRatedList", "dimConditionsList", "condDimensionMap",
          "condContextsList", "contextConditionsList", "uiUserIds", "uiItemIds")


def new_fields(path, share=None):
    f = {"dataPath": path, "fullStat": False, "isHeadline": True, "MaxRate": -1.0, "MinRate": -1.0, "numRatings": 0,
         "scaleDist": javasrc.JMultiset(), "ratingScale": None, "rateMatrix": None, "EmptyContextConditions": None,
         "rates_c": None, "rates_c_count": None, "rates_u_count": None, "rates_i_count": None}
    for name in SHARED:
        if share is not None:
            f[name] = share[name]
        elif name in ("uRatedList", "iRatedList", "dimConditionsList", "condContextsList"):
            f[name] = GuavaMultimap()
        elif name.endswith("Ids") and name not in ("uiUserIds", "uiItemIds"):
            f[name] = javasrc.JBiMap()
        else:
            f[name] = javasrc.JMap()
    return f


def py(v):
    if isinstance(v, Box):
        return v.v
    if isinstance(v, JString):
        return v.s
    return v


def by_inner_id(m):
    """BiMap<String,Integer> -> raw ids in inner-id order (ids are dense 0..n-1 by construction)"""
    inv = {py(v): py(k) for k, v in m.d.items()}
    assert sorted(inv) == list(range(len(inv)))
    return [inv[i] for i in range(len(inv))]


def run_dao(ref, vm, path, share=None):
    src = os.path.join(ref, "src", "carskit", "data", "processor", "DataDAO.java")
    this = javasrc.This(vm, [src], CLASS_MAP)
    this.fields.update(new_fields(path, share))
    mat = this.call("readData", [-1.0])
    F = this.fields
    n_ui, n_ctx = this.call("numUserItems", []), this.call("numContexts", [])
    cells = []
    row_ptr, col_ind, row_data = (mat.fields[n].data for n in ("rowPtr", "colInd", "rowData"))
    for r in range(n_ui):
        for p in range(row_ptr[r], row_ptr[r + 1]):
            cells.append([r, int(col_ind[p]), float(row_data[p]).hex()])
    n_conds = this.call("numConditions", [])
    out = {"users": by_inner_id(F["userIds"]), "items": by_inner_id(F["itemIds"]), "uis": by_inner_id(F["uiIds"]),
           "ctxs": by_inner_id(F["ctxIds"]), "conds": by_inner_id(F["condIds"]), "dims": by_inner_id(F["dimIds"]),
           "cond_dim": [py(F["condDimensionMap"].d[Box(c, "Integer")]) for c in range(n_conds)],
           "empty": [py(x) for x in F["EmptyContextConditions"].items],
           "ui_user": [py(F["uiUserIds"].d[Box(i, "Integer")]) for i in range(n_ui)],
           "ui_item": [py(F["uiItemIds"].d[Box(i, "Integer")]) for i in range(n_ui)],
           "ctx_conds": [[py(x) for x in F["contextConditionsList"].d[Box(c, "Integer")].items] for c in range(n_ctx)],
           "num_ratings": this.call("numRatings", []), "scale": [float(py(x)).hex() for x in F["ratingScale"].items],
           "counts": [this.call(m, []) for m in ("numUsers", "numItems", "numUserItems", "numContexts", "numConditions", "numContextDims")],
           "cells": cells, "statements": this.statements}
    return out, F


def run_transform(ref, vm, f_train, p_train, f_test, p_test, outdir):
    """DataTransformer.setParameters + run() (DataTransformer.java:49-55, 332-396) from source; returns the files it wrote"""
    src = os.path.join(ref, "src", "carskit", "data", "processor", "DataTransformer.java")
    this = javasrc.This(vm, [src], {})
    for name in ("train.csv", "test.csv"):
        if os.path.exists(os.path.join(outdir, name)):
            os.unlink(os.path.join(outdir, name))
    this.call("setParameters", [f_train, p_train, f_test, p_test, outdir])
    this.call("run", [])
    out = {}
    for name in ("train.csv", "test.csv"):
        q = os.path.join(outdir, name)
        if os.path.exists(q):
            out[name] = open(q, newline="").read()
    out["statements"] = this.statements
    return out


def mint_splitter(ref, vm_unused, golden):
    """DataSplitter.splitFolds + getKthFold (DataSplitter.java:68-133) from source; happy.coding.math.Randoms (seed, uniform) and
    Sortor.quickSort from the happy.coding.utils jar's bytecode, SparseMatrix copy / set / reshape from the librec jar's"""
    vm = VM([os.path.join(ref, "lib", "librec-v1.4-alpha.jar"), os.path.join(ref, "lib", "happy.coding.utils-1.2.6.jar")])
    src = os.path.join(ref, "src", "carskit", "data", "processor", "DataSplitter.java")
    cmap = {"SparseMatrix": SM, "Randoms": "happy/coding/math/Randoms", "Sortor": "happy/coding/math/Sortor"}
    cases = []
    for (n_rows, n_cols, n_cells, kfold, seed) in ((6, 5, 17, 5, 1), (9, 4, 23, 3, 42), (4, 3, 4, 10, 7), (12, 6, 50, 5, 20260928)):
        rng = random.Random(seed * 7 + 1)
        cells = {}
        while len(cells) < n_cells:
            cells[(rng.randrange(n_rows), rng.randrange(n_cols))] = float(rng.randrange(1, 6))
        cells = [[r, c, v] for (r, c), v in sorted(cells.items())]
        from oracle.mint_reference_src import sparse
        mat = sparse(vm, n_rows, n_cols, cells)
        this = javasrc.This(vm, [src], cmap)
        this.fields.update({"rateMatrix": mat, "assignMatrix": None, "numFold": 0})
        this.override["debugInfo"] = lambda *a: None
        javasrc.vm_call(vm, None, "happy/coding/math/Randoms", "seed", [javasrc.JLong(seed)], static=True)
        this.call("splitFolds", [kfold])
        am = this.fields["assignMatrix"]
        rp, ci, rd = (am.fields[x].data for x in ("rowPtr", "colInd", "rowData"))
        labels = [int(rd[p]) for r in range(n_rows) for p in range(rp[r], rp[r + 1])]
        folds = []
        for f in range(1, this.fields["numFold"] + 1):
            tr, te = this.call("getKthFold", [f])
            def crs(m):
                a, b, d = (m.fields[x].data for x in ("rowPtr", "colInd", "rowData"))
                return [[r, int(b[p]), float(d[p]).hex()] for r in range(n_rows) for p in range(a[r], a[r + 1])]
            folds.append({"train": crs(tr), "test": crs(te)})
        cases.append({"name": "split_%d_of_%d_seed_%d" % (kfold, n_cells, seed), "n_rows": n_rows, "n_cols": n_cols, "cells": cells,
                      "kfold": kfold, "seed": seed, "num_fold": this.fields["numFold"], "labels": labels, "folds": folds,
                      "statements": this.statements})
        print("splitter", cases[-1]["name"], "numFold", this.fields["numFold"], labels[:12], flush=True)
    return cases


VALIDATE_TEXTS = {
    "binary_plain": "User,Item,Rating,time:na,time:weekend\nu,i,3,1,0\n",
    "binary_digits_10": "User,Item,Rating,time:na,time:weekend\nu,i,3,10,11\n",          # isBinaryNumber looks at decimal digits
    "binary_negative": "User,Item,Rating,time:na,time:weekend\nu,i,3,-5,0\n",             # -5 % 10 = -5, not > 1: "binary"
    "binary_digit_2": "User,Item,Rating,time:na,time:weekend\nu,i,3,1,2\n",               # a 2: compact
    "binary_padded_value": "User,Item,Rating,time:na,time:weekend\nu,i,3, 1,0\n",         # Integer.valueOf does not trim: throws
    "binary_plus_sign": "User,Item,Rating,time:na,time:weekend\nu,i,3,+1,0\n",
    "colon_header_text_value": "User,Item,Rating,time:na,time:weekend\nu,i,3,yes,0\n",    # throws
    "compact_plain": "User,Item,Rating,Time,Location\nu,i,3,Weekend,Home\n",
    "compact_mixed_header": "User,Item,Rating,time:na,Location\nu,i,3,1,Home\n",
    "loose_plain": "user,item,rating,dimension,condition\nu,i,3,Time,Weekend\n",
    "loose_padded_caps": "user,item,rating, Dimension , CONDITION \nu,i,3,Time,Weekend\n",
    "no_context_columns": "User,Item,Rating\nu,i,3\n",                                    # the loop never runs: "binary"
    "short_data_line": "User,Item,Rating,time:na,time:weekend\nu,i,3,1\n",                # sdata[4] out of bounds: throws
    "header_only": "User,Item,Rating,time:na\n",                                            # dataline == null: throws
}


def mint_validate(ref, vm, golden):
    """CARSKit.validateDataFormat (CARSKit.java:177-215) from source"""
    import tempfile
    src = os.path.join(ref, "src", "carskit", "main", "CARSKit.java")
    cases = []
    tmp = os.path.join(tempfile.mkdtemp(prefix="mint_vf_"), "f.csv")

    def run(path):
        this = javasrc.This(vm, [src], {})
        try:
            return {"format": this.call("validateDataFormat", [path]), "statements": this.statements}
        except (RuntimeError, IndexError, AttributeError) as e:
            return {"throws": "%s: %s" % (type(e).__name__, e)}
    for name in ("train_binary.csv", "train_loose.csv", "train_compact.csv", "test_binary.csv", "test_loose.csv", "test_compact.csv"):
        cases.append({"name": name, "file": name, "expect": run(os.path.join(golden, name))})
    for name, text in VALIDATE_TEXTS.items():
        open(tmp, "w", newline="").write(text)
        cases.append({"name": name, "text": text, "expect": run(tmp)})
        print("validate", name, cases[-1]["expect"], flush=True)
    return cases


def gen_loose(rng, n, messy):
    dims = {"Time": ["Weekend", "Weekday", ""], "Location": ["Home", "Cinema", "NA"], "Companion": ["Alone", "Family"]}
    lines = ["user,item,rating,dimension,condition"]
    for _ in range(n):
        u, i, r = rng.randrange(8), rng.randrange(5), rng.randrange(1, 6)
        for d in rng.sample(sorted(dims), rng.randrange(1, 4)):
            lines.append(("%d, tt%d ,%d,%s,%s" if messy else "%d,tt%d,%d,%s,%s") % (u, i, r, d, rng.choice(dims[d])))
    return "\n".join(lines) + "\n"


def gen_compact(rng, n, messy):
    dims = {"Time": ["Weekend", "Weekday", ""], "Location": ["Home", "Cinema", "NA"], "Companion": ["Alone", "Family", " Partner "]}
    lines = ["user,item,rating," + ",".join(" %s" % d if messy else d for d in dims)]
    for _ in range(n):
        lines.append("%d,%stt%d,%d,%s" % (rng.randrange(9), " " if messy and rng.random() < 0.3 else "", rng.randrange(6), rng.randrange(1, 6),
                                          ",".join(rng.choice(v) for v in dims.values())))
    return "\n".join(lines) + "\n"


def mint_transform(ref, vm, golden):
    import tempfile
    cases = []
    td = tempfile.mkdtemp(prefix="mint_tr_") + os.sep
    flag = {"binary": 1, "loose": 2, "compact": 3}
    for kind in ("loose", "compact", "binary"):
        out = run_transform(ref, vm, flag[kind], os.path.join(golden, "train_%s.csv" % kind), -1, None, td)
        cases.append({"name": "sample_%s" % kind, "train_file": "train_%s.csv" % kind, "train_format": flag[kind], "expect": out})
    for tr, te in (("loose", "loose"), ("compact", "compact"), ("binary", "binary"), ("compact", "loose"), ("binary", "compact"),
                   ("loose", "binary")):
        out = run_transform(ref, vm, flag[tr], os.path.join(golden, "train_%s.csv" % tr), flag[te], os.path.join(golden, "test_%s.csv" % te), td)
        cases.append({"name": "sample_%s_with_test_%s" % (tr, te), "train_file": "train_%s.csv" % tr, "train_format": flag[tr],
                      "test_file": "test_%s.csv" % te, "test_format": flag[te], "expect": out})
    for seed, kind, messy in ((11, "loose", False), (12, "loose", True), (13, "compact", False), (14, "compact", True), (21, "loose", True),
                              (22, "compact", True), (23, "loose", False), (24, "compact", False), (25, "compact", True), (26, "loose", True)):
        rng = random.Random(seed)
        a = (gen_loose if kind == "loose" else gen_compact)(rng, 40, messy)
        b = (gen_loose if kind == "loose" else gen_compact)(rng, 15, messy)
        pa, pb = os.path.join(td, "in_a.csv"), os.path.join(td, "in_b.csv")
        open(pa, "w", newline="").write(a)
        open(pb, "w", newline="").write(b)
        out = run_transform(ref, vm, flag[kind], pa, -1, None, td)
        cases.append({"name": "generated_%d_%s" % (seed, kind), "train_text": a, "train_format": flag[kind], "expect": out})
        out = run_transform(ref, vm, flag[kind], pa, flag[kind], pb, td)
        cases.append({"name": "generated_%d_%s_with_test" % (seed, kind), "train_text": a, "train_format": flag[kind], "test_text": b,
                      "test_format": flag[kind], "expect": out})
        print("transform", seed, kind, messy, out["statements"], "statements", flush=True)
    return cases


def gen_file(rng, n_lines, n_users, n_items, dims, messy):
    """a rating file in the binary format (my own generator: data, not reference text)"""
    conds = [(d, c) for d, k in enumerate(dims) for c in range(k)]
    sep = ",\t" if messy else ","
    hdr = ["User", " Item", " Rating"] + [" dim%d:%s" % (d, "na" if c == 0 else "c%d" % c) for d, c in conds]
    lines = [sep.join(hdr) + ("  " if messy else "")]
    for _ in range(n_lines):
        u = "u%d" % rng.randrange(n_users)
        i = "i%d" % rng.randrange(n_items)
        if messy and rng.random() < 0.2:
            u = u + " "
        r = rng.choice(["1", "2", "3.5", "4", "5", "0", " 2 ", "4.0d", "1e0"]) if messy else str(rng.randrange(1, 6))
        bits = []
        for d, k in enumerate(dims):
            on = rng.randrange(k) if (not messy or rng.random() < 0.9) else -1
            bits += [(" 1" if messy else "1") if c == on else "0" for c in range(k)]
        lines.append("%s%s,%s,%s%s" % ("  " if messy else "", u, i, r, "".join("," + b for b in bits)))
    return ("\r\n" if messy else "\n").join(lines) + ("" if messy else "\n")


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    vm = VM(os.path.join(ref, "lib", "librec-v1.4-alpha.jar"))
    golden = os.path.join(ROOT, "tests", "golden")
    cases = []
    # the reference's own sample files: train alone, test alone, test after train with shared maps (the -testset flow)
    tr, F = run_dao(ref, vm, os.path.join(golden, "train_binary.csv"))
    cases.append({"name": "sample_train", "file": "train_binary.csv", "expect": tr})
    te, _ = run_dao(ref, vm, os.path.join(golden, "test_binary.csv"))
    cases.append({"name": "sample_test", "file": "test_binary.csv", "expect": te})
    # the sample test file's header lists other columns than the training file's: with the training DAO's maps (the `-testset` flow,
    # CARSKit.java:335) condIds.put("time:weekend", 0) hits HashBiMap's "value already present" -- the reference throws, so must we
    try:
        run_dao(ref, vm, os.path.join(golden, "test_binary.csv"), share=F)
        raise SystemExit("expected the shared read of a file with another header to throw")
    except RuntimeError as e:
        assert "IllegalArgumentException" in str(e)
        cases.append({"name": "sample_test_after_train", "file": "test_binary.csv", "after_file": "train_binary.csv", "throws": str(e)})
    tmp = os.path.join("/tmp", "mint_dao_%d.csv" % os.getpid())
    more = [(100 + i, bool(i % 2), 20 + 7 * i, 3 + i % 9, 2 + (i * 5) % 11, tuple(2 + (i + d) % 3 for d in range(1 + i % 4))) for i in range(16)]
    for seed, messy, n_lines, nu, ni, dims in [(1, False, 60, 9, 7, (3, 2)), (2, True, 80, 8, 6, (2, 3, 2)), (3, True, 40, 5, 5, (4,)),
                                               (4, False, 120, 20, 10, (2, 2, 2))] + more:
        text = gen_file(random.Random(seed), n_lines, nu, ni, dims, messy)
        with open(tmp, "w", newline="") as fh:
            fh.write(text)
        exp, _ = run_dao(ref, vm, tmp)
        cases.append({"name": "generated_%d%s" % (seed, "_messy" if messy else ""), "text": text, "expect": exp})
        print("generated", seed, messy, exp["counts"], exp["num_ratings"], exp["statements"], "statements", flush=True)
    # a test file read after its training file (one header, as DataTransformer writes them): the test DAO extends the training maps
    for seed, messy in ((5, False), (6, True)):
        text = gen_file(random.Random(seed), 90, 10, 8, (3, 2, 2), messy)
        eol = "\r\n" if messy else "\n"
        lines = text.split(eol)
        if lines[-1] == "":
            lines.pop()
        a = eol.join([lines[0]] + lines[1:61]) + eol
        b = eol.join([lines[0]] + lines[61:]) + eol
        with open(tmp, "w", newline="") as fh:
            fh.write(a)
        exp_a, F = run_dao(ref, vm, tmp)
        with open(tmp, "w", newline="") as fh:
            fh.write(b)
        exp_b, _ = run_dao(ref, vm, tmp, share=F)
        cases.append({"name": "pair_%d%s" % (seed, "_messy" if messy else ""), "after_text": a, "text": b, "expect": exp_b})
        print("pair", seed, exp_a["counts"], "->", exp_b["counts"], flush=True)
    os.unlink(tmp)
    scases = mint_splitter(ref, vm, golden)
    vcases = mint_validate(ref, vm, golden)
    tcases = mint_transform(ref, vm, golden)
    out = os.path.join(golden, "reference_transform.json")
    with open(out, "w") as fh:
        json.dump({"note": "minted by oracle/mint_reference_dao.py: DataTransformer.run of the reference, interpreted from its Java source "
                           "(java.util.HashMap's iteration order simulated: the JDK is not in the reference tree); validate = "
                           "CARSKit.validateDataFormat", "cases": tcases, "validate": vcases, "splitter": scases}, fh,
                  separators=(",", ":"))
    print("wrote", out, os.path.getsize(out), "bytes")
    out = os.path.join(golden, "reference_dao.json")
    with open(out, "w") as fh:
        json.dump({"note": "minted by oracle/mint_reference_dao.py: DataDAO.readData of the reference, interpreted from its Java source; "
                           "doubles as hex", "cases": cases}, fh, separators=(",", ":"))
    print("wrote", out, os.path.getsize(out), "bytes;", [c["expect"]["counts"] for c in cases if "expect" in c][:3])


if __name__ == "__main__":
    main()

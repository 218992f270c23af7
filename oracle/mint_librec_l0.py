#!/usr/bin/env python3
"""Mint tests/golden/librec_l0.json by EXECUTING the reference's vendored librec bytecode (build container only).

    python oracle/mint_librec_l0.py [/root/reference/lib/librec-v1.4-alpha.jar]

The reference has no tests or golden vectors and this image has no JVM, so the L0 layer of the oracle -- the order of
DenseMatrix.rowMult / DenseVector.inner, the CRS layout and iteration order of SparseMatrix, what size() / sum() count, the order
in which init() draws from the RNG -- was known only from a reading of the jar's bytecode.  oracle/jvm is a small class-file
interpreter; this script runs those methods from the jar itself on small seeded inputs and writes inputs + outputs as data
(doubles as C99 hex strings, exact).  tests/test_librec_l0.py then holds the C oracle (and the product's CRS builder) to them.
The jar is read where it lies; nothing of it is copied -- the fixture holds numbers only.  JDK / guava classes the methods call
are host stand-ins written from their specifications (oracle/jvm/interp.py says which): the pin is on librec's own code.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.jvm.interp import VM, GuavaMultimap, GuavaTable, darray, dmatrix, to_list  # noqa: E402

DM, DV, SM = "librec/data/DenseMatrix", "librec/data/DenseVector", "librec/data/SparseMatrix"


def hx(v):
    return float(v).hex()


def dense(vm, rows):
    m = vm.new_object(DM)
    vm.call(DM, "<init>", "([[D)V", [m, dmatrix(rows)])
    return m


def main():
    jar = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/lib/librec-v1.4-alpha.jar"
    vm = VM(jar)
    rng = np.random.default_rng(20260928)
    out = {"source": "librec-v1.4-alpha.jar executed by oracle/jvm (class-file interpreter); doubles are C99 hex strings",
           "row_mult": [], "inner": [], "init_streams": [], "add": [], "sparse": [], "stats": []}

    # ---- DenseMatrix.rowMult / DenseVector.inner: operands of mixed magnitude so that the summation order shows in the last bits
    for k in (1, 2, 3, 7, 10, 64, 70, 128):
        a = rng.standard_normal(k) * 10.0 ** rng.integers(-6, 6, k)
        b = rng.standard_normal(k) * 10.0 ** rng.integers(-6, 6, k)
        pad_a, pad_b = rng.standard_normal(k), rng.standard_normal(k)      # other rows: the row index must be honoured
        A, B = dense(vm, [pad_a, a]), dense(vm, [b, pad_b])
        r = vm.call(DM, "rowMult", "(L%s;IL%s;I)D" % (DM, DM), [A, 1, B, 0])
        out["row_mult"].append({"k": k, "a": [hx(x) for x in a], "b": [hx(x) for x in b], "result": hx(r)})
        va, vb = vm.new_object(DV), vm.new_object(DV)
        vm.call(DV, "<init>", "([D)V", [va, darray(a)])
        vm.call(DV, "<init>", "([D)V", [vb, darray(b)])
        out["inner"].append({"k": k, "a": [hx(x) for x in a], "b": [hx(x) for x in b],
                             "result": hx(vm.call(DV, "inner", "(L%s;)D" % DV, [va, vb]))})

    # ---- init(): the order in which the containers of initModel() draw from ONE librec.util.Randoms stream
    # (IterativeRecommender.java:235-244 P then Q gaussian(0, 0.1); CAMF_CI.java:55-60 userBias gaussian, icBias uniform(0,1))
    for seed, (nu, ni, k, nc) in ((42, (3, 4, 5, 3)), (7, (2, 2, 1, 2)), (20260927, (4, 3, 10, 6))):
        vm.call("librec/util/Randoms", "seed", "(J)V", [seed])
        P, Q, icb = vm.new_object(DM), vm.new_object(DM), vm.new_object(DM)
        vm.call(DM, "<init>", "(II)V", [P, nu, k])
        vm.call(DM, "<init>", "(II)V", [Q, ni, k])
        vm.call(DM, "<init>", "(II)V", [icb, ni, nc])
        ub, cb = vm.new_object(DV), vm.new_object(DV)
        vm.call(DV, "<init>", "(I)V", [ub, nu])
        vm.call(DV, "<init>", "(I)V", [cb, nc])
        vm.call(DM, "init", "(DD)V", [P, 0.0, 0.1])
        vm.call(DM, "init", "(DD)V", [Q, 0.0, 0.1])
        vm.call(DV, "init", "(DD)V", [ub, 0.0, 0.1])
        vm.call(DM, "init", "()V", [icb])                     # uniform(0, 1)
        vm.call(DV, "init", "()V", [cb])                      # DenseVector.init(): uniform(0, 1)
        flat = lambda m: [hx(x) for row in to_list(m.fields["data"]) for x in row]
        out["init_streams"].append({"seed": seed, "n_users": nu, "n_items": ni, "k": k, "n_conds": nc,
                                    "order": ["P gaussian(0,0.1)", "Q gaussian(0,0.1)", "userBias gaussian(0,0.1)", "icBias uniform(0,1)",
                                              "condVector uniform(0,1)"],
                                    "P": flat(P), "Q": flat(Q), "userBias": [hx(x) for x in to_list(ub.fields["data"])],
                                    "icBias": flat(icb), "condVector": [hx(x) for x in to_list(cb.fields["data"])]})

    # ---- DenseMatrix.add(i, j, v) / DenseVector.add(i, v): data[i][j] += v
    m = dense(vm, [[0.1, 0.2], [0.3, 0.4]])
    vm.call(DM, "add", "(IID)V", [m, 1, 0, 1e-17])
    vm.call(DM, "add", "(IID)V", [m, 0, 1, 0.7])
    v = vm.new_object(DV)
    vm.call(DV, "<init>", "([D)V", [v, darray([1.0, 2.0])])
    vm.call(DV, "add", "(ID)V", [v, 1, 0.1])
    out["add"].append({"matrix_after": [[hx(x) for x in r] for r in to_list(m.fields["data"])], "vector_after": [hx(x) for x in to_list(v.fields["data"])]})

    # ---- SparseMatrix: construct from a (row, column) -> value table, iterate, size(), sum(), reshape()
    for case in range(6):
        n_rows, n_cols = int(rng.integers(1, 9)), int(rng.integers(1, 7))
        n_put = int(rng.integers(0, 2 * n_rows * n_cols))
        puts = []
        t, cm = GuavaTable(), GuavaMultimap()
        for _ in range(n_put):
            r, c = int(rng.integers(0, n_rows)), int(rng.integers(0, n_cols))
            val = float(rng.integers(0, 6))                    # ratings 0..5: explicit zeros and repeated cells (last write wins) occur
            puts.append([r, c, val])
            t.put(r, c, val)
            cm.put(c, r)
        S = vm.new_object(SM)
        vm.call(SM, "<init>", "(IILcom/google/common/collect/Table;Lcom/google/common/collect/Multimap;)V", [S, n_rows, n_cols, t, cm])
        it = vm.call(SM, "iterator", "()Ljava/util/Iterator;", [S])
        entries = []
        while vm.invoke("interface", "java/util/Iterator", "hasNext", "()Z", [it]):
            e = vm.invoke("interface", "java/util/Iterator", "next", "()Ljava/lang/Object;", [it])
            entries.append([vm.invoke("interface", "librec/data/MatrixEntry", "row", "()I", [e]),
                            vm.invoke("interface", "librec/data/MatrixEntry", "column", "()I", [e]),
                            vm.invoke("interface", "librec/data/MatrixEntry", "get", "()D", [e])])
        rec = {"n_rows": n_rows, "n_cols": n_cols, "puts": puts,
               "rowPtr": to_list(S.fields["rowPtr"]), "colInd": to_list(S.fields["colInd"]), "rowData": to_list(S.fields["rowData"]),
               "colPtr": to_list(S.fields["colPtr"]), "rowInd": to_list(S.fields["rowInd"]), "colData": to_list(S.fields["colData"]),
               "iterator": entries, "size": vm.call(SM, "size", "()I", [S]), "sum": hx(vm.call(SM, "sum", "()D", [S]))}
        # carskit.data.structure.SparseMatrix.getGlobalAvg (src, :49-56) = sum() / size(): the two jar methods it is made of
        rec["global_avg"] = hx(float.fromhex(rec["sum"]) / rec["size"]) if rec["size"] else None
        vm.call(SM, "reshape", "(L%s;)V" % SM, [S])            # DataSplitter.java:88-89,162-163: drops the zero entries
        rec["after_reshape"] = {"rowPtr": to_list(S.fields["rowPtr"]), "colInd": to_list(S.fields["colInd"]), "rowData": to_list(S.fields["rowData"])}
        out["sparse"].append(rec)

    # ---- Stats.sum / Stats.mean over double[]
    for n in (1, 5, 33):
        x = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4, n)
        out["stats"].append({"x": [hx(v_) for v_ in x], "sum": hx(vm.call("librec/util/Stats", "sum", "([D)D", [darray(x)])),
                             "mean": hx(vm.call("librec/util/Stats", "mean", "([D)D", [darray(x)]))})

    out["bytecode_steps"] = vm.steps
    path = os.path.join(ROOT, "tests", "golden", "librec_l0.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote %s (%d bytecode instructions interpreted)" % (path, vm.steps))


if __name__ == "__main__":
    main()

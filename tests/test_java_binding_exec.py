"""The Java side of the boundary, EXECUTED (there is no JDK here, so until round 3 java/carskit/alg/gpu/*.java was checked as text only).

oracle/check_java_binding.py puts each `X_GPU.java` drop-in in front of the reference's own class chain and interprets it with
oracle/jvm/javasrc.py: `buildModel()` -> `GpuSupport.buildModel(this)` -> marshalling, the epoch loop with the reference's unchanged
`isConverged()`, copy-back; the `NativeMF` natives land in a stand-in that drives the order-exact CPU oracle with the arguments the C ABI
would get.  The bar is bit-identity with the reference's own `buildModel()` (tests/golden/reference_src.json).

The committed verdict (tests/golden/java_binding_check.json) is checked everywhere; the live re-run needs the reference tree
(/root/reference: present in the build container, absent on the GPU box) and is skipped without it."""
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ALL = json.load(open(os.path.join(ROOT, "tests", "golden", "java_binding_check.json")))
CHECK = _ALL["models"]
REF = os.environ.get("CARSKIT_REFERENCE", "/root/reference")


def test_committed_verdict_covers_every_sgd_drop_in_and_is_bit_identical():
    assert set(CHECK) == {"BiasedMF", "PMF", "CAMF_C", "CAMF_CI", "CAMF_CU", "CAMF_CUCI", "SVD++", "CAMF_ICS", "CAMF_LCS", "CAMF_MCS", "FM"}
    for model, rec in CHECK.items():
        assert all(rec["bit_identical"].values()), (model, rec["bit_identical"])
        assert ({"w0", "w", "V"} if model == "FM" else {"P", "Q", "epoch_loss", "epoch_lrate"}) <= set(rec["bit_identical"])
        assert os.path.exists(os.path.join(ROOT, "java", "carskit", "alg", "gpu", rec["drop_in"] + ".java"))


def test_native_call_sequence_is_what_the_jni_shim_exports():
    jni = open(os.path.join(ROOT, "jni", "carskit_jni.cpp")).read()
    native_mf = open(os.path.join(ROOT, "java", "carskit", "alg", "gpu", "NativeMF.java")).read()
    for model, rec in CHECK.items():
        calls = rec["native_calls"]
        for name in set(calls):
            assert re.search(r"Java_carskit_alg_gpu_NativeMF_%s\b" % name, jni), name
            assert re.search(r"public static native \S+ %s\(" % name, native_mf), name
        if model == "FM":
            assert calls == ["fmCreate", "fmSetHparams", "fmSetRatingsCsr", "fmSetModel", "fmTrain", "fmGetModel", "fmDestroy"]
            continue
        assert calls[0] == "create" and calls[-1] == "destroy", model
        for name in set(calls):
            assert re.search(r"Java_carskit_alg_gpu_NativeMF_%s\b" % name, jni), name
            assert re.search(r"public static native \S+ %s\(" % name, native_mf), name
        first_epoch = calls.index("trainEpoch")
        assert "setHparams" in calls[:first_epoch] and any(c.startswith("setRatings") for c in calls[:first_epoch])
        assert "setMatrix" in calls[:first_epoch]                       # copy-in before the first epoch
        last_epoch = len(calls) - 1 - calls[::-1].index("trainEpoch")
        assert "getMatrix" in calls[last_epoch:]                        # copy-back after the last one
        assert not any(c.startswith("set") for c in calls[first_epoch:])


def test_committed_early_stop_run_is_bit_identical_and_scores_the_live_model():
    """`--early-stop RMSE`: the reference's isConverged() calls evalRatings() every epoch; the drop-in answers from the native model"""
    rec = _ALL["early_stop_rmse"]
    assert all(rec["bit_identical"].values()), rec["bit_identical"]
    calls = rec["native_calls"]
    assert "setEvalRatings" in calls[:calls.index("trainEpoch")]
    assert calls.count("evalResident") == calls.count("trainEpoch")          # one score per epoch
    for a, b in zip(calls, calls[1:]):
        assert not (a == "trainEpoch" and b == "trainEpoch")                 # ... taken between the epochs, not after the loop


def test_committed_sharded_run_uses_the_group_natives_only():
    """-Dcarskit.shards=2: GpuSupport.buildModelSharded, containers routed through Dev.ofGroup's negative handles"""
    rec = _ALL["shards_2"]
    assert all(rec["bit_identical"].values()), rec["bit_identical"]
    calls = rec["native_calls"]
    assert calls[0] == "groupCreate" and calls[-1] == "groupDestroy" and all(c.startswith("group") for c in calls)
    first = calls.index("groupTrainEpoch")
    assert {"groupSetHparams", "groupSetRatingsCsr", "groupSetMatrix", "groupSetLrScale"} <= set(calls[:first])
    jni = open(os.path.join(ROOT, "jni", "carskit_jni.cpp")).read()
    for name in set(calls):
        assert re.search(r"Java_carskit_alg_gpu_NativeMF_%s\b" % name, jni), name


def test_committed_gpu_ranking_runs_give_the_reference_measures():
    """-Dcarskit.gpu.rank=true: evalRankings() of every drop-in through GpuSupport.evalRankings -> NativeMF.evalRankings"""
    assert set(_ALL["rank_on_gpu"]) == {"BiasedMF", "PMF", "CAMF_C", "CAMF_CI", "CAMF_CU", "CAMF_CUCI", "SVD++", "CAMF_ICS", "CAMF_LCS", "CAMF_MCS",
                                         "FM"}
    for model, rec in _ALL["rank_on_gpu"].items():
        assert len(rec["same_measures"]) == 21 and all(rec["same_measures"].values()), model
        calls = rec["native_calls"]
        if model == "FM":
            assert calls == ["fmCreate", "fmSetHparams", "fmSetRatingsCsr", "fmSetModel", "fmEvalRankings", "fmDestroy"]
            continue
        tail = calls[len(calls) - 1 - calls[::-1].index("create"):]          # the evaluation's own handle
        assert tail[0] == "create" and tail[-1] == "destroy" and tail[-2] == "evalRankings", (model, tail)
        assert "setHparams" in tail and any(c.startswith("setRatings") for c in tail) and "setMatrix" in tail
        assert "trainEpoch" not in tail


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "carskit")), reason="needs the reference tree (build container only)")
def test_drop_ins_execute_bit_identically_to_the_reference_buildmodel():
    from oracle import check_java_binding as chk
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_src.json")))["cases"]
    seen = set()
    for case in cases:
        if case["model"] in seen:
            continue
        seen.add(case["model"])
        calls, same, stmts = chk.check(REF, case)
        assert all(same.values()), (case["model"], same)
        assert calls == CHECK[case["model"]]["native_calls"], case["model"]
        assert stmts > 100
    assert len(seen) == 10
    fm = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_src.json")))["fm_cases"][0]
    calls, same, _ = chk.check_fm(REF, fm)
    assert all(same.values()) and calls == CHECK["FM"]["native_calls"]
    calls, same = chk.check_early_stop(REF, [c for c in cases if c["model"] == "CAMF_CU"][0])
    assert all(same.values()) and calls == _ALL["early_stop_rmse"]["native_calls"]
    rank_case = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_rank.json")))["cases"][0]
    calls, same = chk.check_rank(REF, rank_case)
    assert all(same.values()) and calls == _ALL["rank_on_gpu"][rank_case["model"]]["native_calls"]
    calls, same = chk.check_fm_rank(REF, json.load(open(os.path.join(ROOT, "tests", "golden", "reference_rank.json")))["fm_cases"][0])
    assert all(same.values()) and calls == _ALL["rank_on_gpu"]["FM"]["native_calls"]
    calls, same = chk.check_group(REF, [c for c in cases if c["model"] == "CAMF_CI"][0])
    assert all(same.values()) and calls == _ALL["shards_2"]["native_calls"]

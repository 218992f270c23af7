"""BASELINE configs C1 and C2 as GPU parity cases.
C1: BiasedMF k=10 on DePaulMovie through setting.conf and the driver, engine = libcarskit_mi355x (fp32 state),
    against the committed golden MAE/RMSE minted by the CPU oracle (tests/golden/golden_c1_depaul_biasedmf.json).
C2: CAMF_C k=64 fp32 on a Frappe-SHAPED synthetic set (957 users, 4 082 items, 8 dimensions / 343 conditions,
    96 203 ratings; the real Frappe file may not be redistributed), serial schedule, against the oracle."""
import json
import os
import shutil

import numpy as np
import pytest

from carskit_amd import capi, synth
from tests.hostmirror import main, recommender
from tests import util

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_c1_biasedmf_depaul_on_gpu_matches_golden(tmp_path):
    shutil.copyfile(os.path.join(GOLDEN, "depaul_ratings_compact.csv"), tmp_path / "ratings.txt")
    conf = open(os.path.join(GOLDEN, "depaul_setting.conf")).read().replace("PLACEHOLDER_SET_BY_TEST", str(tmp_path / "ratings.txt"))
    (tmp_path / "setting.conf").write_text(conf)
    lines = []
    avg, algos, _ = main.run(str(tmp_path / "setting.conf"), log=lines.append, conf_overrides={"num_iters": 30})
    want = json.load(open(os.path.join(GOLDEN, "golden_c1_depaul_biasedmf.json")))
    for a, w in zip(algos, want["folds"]):
        assert a.testMatrix.n == w["n_test"]
        assert abs(a.measures["RMSE"] - w["RMSE"]) <= 1e-5 and abs(a.measures["MAE"] - w["MAE"]) <= 1e-5
    assert abs(avg["RMSE"] - want["avg_RMSE"]) <= 1e-5
    assert lines[-1].startswith("Final Results by BiasedMF, MAE: ")
    # fp64 + strict: the driver's numbers are the oracle's, digit for digit
    avg64, algos64, _ = main.run(str(tmp_path / "setting.conf"), log=lambda *a: None,
                                 conf_overrides={"num_iters": 30, "flags": capi.FLAG_STATE_F64 | capi.FLAG_STRICT |
                                                 capi.FLAG_SCHED_SERIAL})
    for a, w in zip(algos64, want["folds"]):
        assert abs(a.measures["RMSE"] - w["RMSE"]) <= 1e-12 and abs(a.measures["MAE"] - w["MAE"]) <= 1e-12


def test_item_ranking_depaul_on_gpu_matches_golden(tmp_path):
    """item.ranking=on through the driver: CAMF_CU trained and ranked on the GPU vs the oracle-minted golden.
    fp64+strict: bit-identical model -> identical lists -> measures to 1e-12; fp32 default: within 0.02."""
    from tests.test_host_layer import _ranking_conf
    conf = _ranking_conf(tmp_path)
    want = json.load(open(os.path.join(GOLDEN, "golden_depaul_camf_cu_ranking.json")))
    lines = []
    _, algos, _ = main.run(conf, log=lines.append, conf_overrides={"num_iters": 20, "flags": capi.FLAG_STATE_F64 | capi.FLAG_STRICT})
    assert lines[-1].startswith("Final Results by CAMF_CU, Pre5: ")
    for a, w in zip(algos, want["folds"]):
        for m, v in w.items():
            assert abs(a.measures[m] - v) <= 1e-12, (m, a.measures[m], v)
    _, algos32, _ = main.run(conf, log=lambda *a: None, conf_overrides={"num_iters": 20})
    for a, w in zip(algos32, want["folds"]):
        for m, v in w.items():
            assert abs(a.measures[m] - v) <= 0.02, (m, a.measures[m], v)


def test_early_stop_rmse_uses_the_resident_test_set(tmp_path):
    """`--early-stop RMSE`: the test tuples stay on the device and are evaluated after every epoch
    (IterativeRecommender.java:156-161).  Same stopping iteration and measures as the oracle-driven host loop, in
    fp64/strict mode to 1e-12; resident and per-call evaluation agree exactly."""
    import re
    import subprocess
    from tests.test_host_layer import EXE, _depaul_conf
    conf = _depaul_conf(tmp_path)
    txt = open(conf).read().replace("--test-view all", "--test-view all --early-stop RMSE")
    open(conf, "w").write(txt)
    over = {"num_iters": 40}
    _, ref_algos, _ = main.run(conf, engine_factory=util.OracleEngine, log=lambda *a: None, conf_overrides=over)
    _, algos, _ = main.run(conf, log=lambda *a: None,
                           conf_overrides=dict(over, flags=capi.FLAG_STATE_F64 | capi.FLAG_STRICT | capi.FLAG_SCHED_SERIAL))
    for a, b in zip(algos, ref_algos):
        assert a.conf.early_stop == "RMSE" and getattr(a.engine, "eval_resident_ready", False)
        assert len(a.losses) == len(b.losses)                     # stopped at the same iteration
        assert abs(a.measures["RMSE"] - b.measures["RMSE"]) <= 1e-12 and abs(a.measure - b.measure) <= 1e-12
    a = algos[0]
    t = a.testMatrix
    direct = a.engine.inst.eval_ratings(t.u, t.j, None, t.r, a.minRate, a.maxRate)
    assert direct == a.engine.eval_resident(a.minRate, a.maxRate)
    # the C++ host takes the same path
    flags = capi.FLAG_STATE_F64 | capi.FLAG_STRICT | capi.FLAG_SCHED_SERIAL
    p = subprocess.run([EXE, "-c", conf, "--iters", "40", "--flags", str(flags), "--precise"], capture_output=True, text=True)
    assert p.returncode == 0 and re.search(r"PRECISE BiasedMF folds=5 MAE=(\S+) RMSE=(\S+)", p.stdout), p.stderr


def frappe_shaped(seed=7):
    """957 users x 4 082 items, 8 context dimensions with Frappe's cardinalities, ~96K ratings."""
    rng = np.random.default_rng(seed)
    n, dims = 96203, [7, 7, 2, 3, 2, 9, 80, 233]
    base = synth.generate(957, 4082, 0, 1, n, seed=seed, item_zipf=1.05)
    m = base.n
    conds = np.stack([rng.integers(0, k, m) for k in dims], axis=1)
    offs = np.concatenate([[0], np.cumsum(dims)[:-1]])
    rows = conds + offs
    keys, first, inv = np.unique(rows, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty(len(keys), np.int64)
    rank[order] = np.arange(len(keys))
    ctx = rank[inv.reshape(-1)].astype(np.int32)
    table = keys[order]
    ctx_ptr = (np.arange(len(table) + 1) * len(dims)).astype(np.int32)
    r = np.clip(np.rint(np.log1p(rng.pareto(1.2, m) * 3)), 1, 9).astype(np.float64)   # heavy-tailed usage counts, log scale
    return synth.RatingData(base.n_users, base.n_items, int(sum(dims)), len(dims), base.u, base.j, ctx, r, ctx_ptr,
                            table.reshape(-1).astype(np.int32), 1.0, 9.0)


def test_c2_camf_c_k64_frappe_shaped():
    data = frappe_shaped()
    assert data.n_conds == 343 and data.n_dims == 8
    train, test = synth.split(data, 0.2)
    conf = recommender.Conf(num_factors=64, num_iters=15, init_lrate=util.LR, bold_driver=True, regU=util.REG,
                            regI=util.REG, regB=util.REG, regC=util.REGC, verbose=False)
    gpu = recommender.CAMF_C(train, test, -1, conf)
    cpu = recommender.CAMF_C(train, test, -1, conf, engine_factory=util.OracleEngine)
    mg, mc = gpu.execute(), cpu.execute()
    assert gpu.lrates == cpu.lrates                               # same bold-driver decisions
    np.testing.assert_allclose(gpu.losses, cpu.losses, rtol=2e-5)
    assert abs(mg["RMSE"] - mc["RMSE"]) <= 1e-5 and abs(mg["MAE"] - mc["MAE"]) <= 1e-5   # north_star fp32 bar
    assert gpu.engine.inst.schedule_info()["kind"] == "serial"


def test_cpp_host_driver_c1_matches_oracle(tmp_path):
    """The C++ host (carskit_amd/bin/carskit-mi355x: setting.conf driver + Recommender classes over the C ABI only) on
    BASELINE config C1.  In fp64 / strict / serial mode every number it prints must be the oracle's for the same folds
    and the same java.util.Random init stream."""
    import re
    import subprocess
    from tests.test_host_layer import EXE, _depaul_conf, expected_from_oracle
    conf = _depaul_conf(tmp_path)
    flags = capi.FLAG_STATE_F64 | capi.FLAG_STRICT | capi.FLAG_SCHED_SERIAL
    p = subprocess.run([EXE, "-c", conf, "--iters", "20", "--flags", str(flags), "--precise"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    m = re.search(r"PRECISE BiasedMF folds=5 MAE=(\S+) RMSE=(\S+)", p.stdout)
    assert m, p.stdout[-500:]
    want = expected_from_oracle(conf, "biasedmf", 20)
    assert abs(float(m.group(1)) - want["MAE"]) <= 1e-12 and abs(float(m.group(2)) - want["RMSE"]) <= 1e-12
    assert re.search(r"Final Results by BiasedMF, MAE: %.6f, RMSE: %.6f, NAME: " % (want["MAE"], want["RMSE"]), p.stdout)
    # default fp32 state: the north_star tolerance
    p32 = subprocess.run([EXE, "-c", conf, "--iters", "20", "--precise"], capture_output=True, text=True)
    m32 = re.search(r"PRECISE BiasedMF folds=5 MAE=(\S+) RMSE=(\S+)", p32.stdout)
    assert abs(float(m32.group(1)) - want["MAE"]) <= 1e-5 and abs(float(m32.group(2)) - want["RMSE"]) <= 1e-5
    # a contextual model through the same driver
    txt = open(conf).read().replace("recommender=biasedmf", "recommender=camf_cu")
    open(conf, "w").write(txt)
    pc = subprocess.run([EXE, "-c", conf, "--iters", "10", "--flags", str(flags), "--precise"], capture_output=True, text=True)
    mc = re.search(r"PRECISE CAMF_CU folds=5 MAE=(\S+) RMSE=(\S+)", pc.stdout)
    assert mc, pc.stdout[-300:] + pc.stderr
    wc = expected_from_oracle(conf, "camf_cu", 10)
    assert abs(float(mc.group(1)) - wc["MAE"]) <= 1e-12 and abs(float(mc.group(2)) - wc["RMSE"]) <= 1e-12


def test_cpp_host_save_model_then_load_model(tmp_path):
    """`output.setup ... --save-model` through the C++ host (Recommender.java:240,364-365; IterativeRecommender.java:249-270): one
    model file per fold appears under <workspace>/<algo>/; a second run with --load-model evaluates those files instead of
    training (the reference's loadModel() branch, Recommender.java:332-338) and prints the identical measures -- context tables
    included, which the reference's own saveModel() forgets."""
    import glob
    import re
    import subprocess
    from tests.test_host_layer import EXE, _depaul_conf
    conf = _depaul_conf(tmp_path)
    txt = open(conf).read().replace("recommender=biasedmf", "recommender=camf_ci").replace("-verbose off", "-verbose off --save-model")
    open(conf, "w").write(txt)
    flags = capi.FLAG_STATE_F64 | capi.FLAG_STRICT
    p = subprocess.run([EXE, "-c", conf, "--iters", "8", "--flags", str(flags), "--precise"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    m = re.search(r"PRECISE CAMF_CI folds=5 MAE=(\S+) RMSE=(\S+)", p.stdout)
    assert m, p.stdout[-500:]
    files = sorted(glob.glob(str(tmp_path / "CARSKit.Workspace" / "CAMF_CI" / "model fold [[]*[]].cmi")))
    assert len(files) == 5, files
    assert "Learned models are saved to folder" in p.stdout
    q = subprocess.run([EXE, "-c", conf, "--iters", "8", "--flags", str(flags), "--precise", "--load-model"], capture_output=True, text=True)
    assert q.returncode == 0, q.stderr
    assert "A recommender model is loaded from" in q.stdout and " iter 1:" not in q.stdout
    m2 = re.search(r"PRECISE CAMF_CI folds=5 MAE=(\S+) RMSE=(\S+)", q.stdout)
    assert m2 and m2.groups() == m.groups(), (m.groups(), m2 and m2.groups())


def test_cpp_host_driver_item_ranking(tmp_path):
    """item.ranking=on through the C++ host: fp64/strict numbers equal the oracle's (training + ranking) for the same
    folds and init stream; the printed line has the reference's layout."""
    import re
    import subprocess
    from tests.test_host_layer import EXE, _ranking_conf, expected_from_oracle
    conf = _ranking_conf(tmp_path, "camf_ci")
    flags = capi.FLAG_STATE_F64 | capi.FLAG_STRICT
    p = subprocess.run([EXE, "-c", conf, "--iters", "10", "--flags", str(flags), "--precise"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    m = re.search(r"PRECISE CAMF_CI folds=5 Pre10=(\S+) Rec10=(\S+) AUC10=(\S+) MAP10=(\S+) NDCG10=(\S+) MRR10=(\S+)", p.stdout)
    assert m, p.stdout[-500:]
    want = expected_from_oracle(conf, "camf_ci", 10)
    for g, name in zip(m.groups(), ("Pre10", "Rec10", "AUC10", "MAP10", "NDCG10", "MRR10")):
        assert abs(float(g) - want[name]) <= 1e-12, (name, g, want[name])
    assert ("Final Results by CAMF_CI, Pre5: %.6f,Pre10: %.6f, Rec5: %.6f" % (want["Pre5"], want["Pre10"], want["Rec5"])) in p.stdout


def test_cpp_host_driver_parallel_folds_and_fm(tmp_path):
    """`cv -p on` (one host thread per fold, as in the reference) gives the same numbers as `-p off`; and FM through the
    C++ driver matches the dense FM oracle on the same folds and init stream."""
    import re
    import subprocess
    from carskit_amd import dao
    from tests.hostmirror import splitter
    from oracle import oracle_c
    from tests.test_host_layer import EXE, _depaul_conf
    conf = _depaul_conf(tmp_path)
    flags = capi.FLAG_STATE_F64 | capi.FLAG_STRICT | capi.FLAG_SCHED_SERIAL
    txt = open(conf).read().replace("recommender=biasedmf", "recommender=camf_ci")
    open(conf, "w").write(txt)
    off = subprocess.run([EXE, "-c", conf, "--iters", "8", "--flags", str(flags), "--precise"], capture_output=True, text=True)
    open(conf, "w").write(txt.replace("-p off", "-p on"))
    on = subprocess.run([EXE, "-c", conf, "--iters", "8", "--flags", str(flags), "--precise"], capture_output=True, text=True)
    a = re.search(r"PRECISE CAMF_CI folds=5 MAE=(\S+) RMSE=(\S+)", off.stdout)
    b = re.search(r"PRECISE CAMF_CI folds=5 MAE=(\S+) RMSE=(\S+)", on.stdout)
    assert a and b and a.groups() == b.groups(), (off.stderr, on.stderr)
    # FM
    open(conf, "w").write(txt.replace("recommender=camf_ci", "recommender=fm"))
    fm = subprocess.run([EXE, "-c", conf, "--iters", "3", "--precise"], capture_output=True, text=True)
    m = re.search(r"PRECISE FM folds=5 MAE=(\S+) RMSE=(\S+)", fm.stdout)
    assert m, fm.stdout[-300:] + fm.stderr
    d = dao.DataDAO(os.path.join(os.path.dirname(conf), "CARSKit.Workspace", "train.csv"))
    data = d.rating_data()
    labels, k = splitter.split_folds(data.n, 5, 1)
    reglw, reglf = synth.java_float(0.01), synth.java_float(0.02)
    maes, rmses = [], []
    for f in range(1, k + 1):
        tr, te = splitter.kth_fold(data, labels, f)
        g = oracle_c.JRandom(1)
        p = tr.n_users + tr.n_items + tr.n_conds
        w = g.uniform((p,))
        V = g.gaussian((p, 10))
        orc = oracle_c.FMOracle(10, tr.n_users, tr.n_items, tr.n_conds, max(1, tr.n_dims), tr.u, tr.j, tr.ctx, tr.r, 0.0, w, V,
                                reglw, reglf)
        orc.init()
        for _ in range(3):
            orc.sweep()
        pred = np.clip([orc.predict(int(u), int(j), int(c)) for u, j, c in zip(te.u, te.j, te.ctx)], 1.0, 5.0)
        err = np.abs(te.r - pred)
        maes.append(err.mean())
        rmses.append(np.sqrt((err * err).mean()))
    assert abs(float(m.group(1)) - np.mean(maes)) <= 1e-9 and abs(float(m.group(2)) - np.mean(rmses)) <= 1e-9


@pytest.mark.parametrize("algo,name", [("svd++", "SVD++"), ("camf_ics", "CAMF_ICS"), ("camf_lcs -f 6", "CAMF_LCS"), ("camf_mcs", "CAMF_MCS")])
def test_n1_recommenders_through_both_hosts_strict_bit_exact(tmp_path, algo, name):
    """SURVEY 8(f) N1 done-criterion: `recommender=svd++|camf_ics|camf_lcs|camf_mcs` through the Python host and the C++ host,
    fp64 + strict: per-fold losses bit-identical to the oracle-driven run and the evaluation measures equal to 1e-12 (DePaulMovie,
    5-fold CV, ':na' conditions from the compact->binary transformer as EmptyContextConditions)."""
    import re
    import subprocess
    from tests.test_host_layer import EXE, _depaul_conf
    conf = _depaul_conf(tmp_path)
    txt = open(conf).read().replace("recommender=biasedmf", "recommender=" + algo).replace("learn.rate=2e-2", "learn.rate=2e-3")
    open(conf, "w").write(txt)
    flags = capi.FLAG_STATE_F64 | capi.FLAG_STRICT | capi.FLAG_SCHED_SERIAL
    _, ref, _ = main.run(conf, engine_factory=util.OracleEngine, log=lambda *a: None, conf_overrides={"num_iters": 6})
    _, gpu, _ = main.run(conf, log=lambda *a: None, conf_overrides={"num_iters": 6, "flags": flags})
    keys = ("MAE", "RMSE") if name == "SVD++" else ("Pre10", "Rec10", "AUC10", "MAP10", "NDCG10", "MRR10")
    for a, b in zip(gpu, ref):
        assert a.losses == b.losses and a.lrates == b.lrates                    # bit-identical epochs
        for n_, arr in a.state.items():
            assert np.array_equal(arr, b.state[n_].reshape(arr.shape)), n_
        for m in keys:
            assert abs(a.measures[m] - b.measures[m]) <= 1e-12, m
    # the C++ host (its own java.util.Random init stream, reproduced by expected_from_oracle): the oracle's numbers, to 1e-12
    from tests.test_host_layer import expected_from_oracle
    p64 = subprocess.run([EXE, "-c", conf, "--iters", "6", "--flags", str(flags), "--precise"], capture_output=True, text=True)
    assert p64.returncode == 0, p64.stderr
    want = expected_from_oracle(conf, algo.split()[0], 6)
    if name == "SVD++":
        m64 = re.search(r"PRECISE SVD\+\+ folds=5 MAE=(\S+) RMSE=(\S+)", p64.stdout)
        assert m64, p64.stdout[-400:] + p64.stderr
        assert abs(float(m64.group(1)) - want["MAE"]) <= 1e-12 and abs(float(m64.group(2)) - want["RMSE"]) <= 1e-12
        assert "Final Results by SVD++, MAE: " in p64.stdout
    else:
        m64 = re.search(r"PRECISE %s folds=5 Pre10=(\S+) Rec10=(\S+) AUC10=(\S+) MAP10=(\S+) NDCG10=(\S+) MRR10=(\S+)" % name, p64.stdout)
        assert m64, p64.stdout[-400:] + p64.stderr
        for g_, key in zip(m64.groups(), ("Pre10", "Rec10", "AUC10", "MAP10", "NDCG10", "MRR10")):
            assert abs(float(g_) - want[key]) <= 1e-12, (key, g_, want[key])
        assert ("Final Results by %s, Pre5: " % name) in p64.stdout

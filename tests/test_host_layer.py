"""Host layer above the C ABI (config parser, java.util.Random stream, fold splitter, recommender classes, driver)
on CPU.  The compute engine is injected (the oracle) -- the product's only engine is the GPU library."""
import json
import os
import shutil

import numpy as np
import pytest

from carskit_amd import synth
from tests.hostmirror import config, javarand, main, recommender, splitter
from oracle import oracle_np
from tests import util

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_line_configer_grammar():
    lc = config.LineConfiger("2e-2 -max -1 -bold-driver")
    assert lc.get_main_param() == "2e-2" and lc.contains("-bold-driver")
    assert lc.get_float("-max") == -1.0                       # "-1" is numeric: a value, not a new key
    assert lc.get_float("-decay", -1.0) == -1.0
    assert config.java_float(lc.get_main_param()) == 0.019999999552965164   # Java float 0.02f promoted
    lc = config.LineConfiger("0.0001 -c 0.001")
    assert config.java_float(lc.get_main_param()) == 9.999999747378752e-05
    assert lc.get_float("-c") == float(np.float32(0.001)) and lc.get_float("-u", 7.0) == 7.0
    lc = config.LineConfiger("cv -k 5 -p on --rand-seed 1 --test-view all")
    assert lc.get_main_param() == "cv" and lc.get_int("-k") == 5 and lc.is_on("-p") and lc.get_long("--rand-seed") == 1
    assert lc.get_string("--test-view") == "all" and lc.get_string("--early-stop") is None
    lc = config.LineConfiger("-folder CARSKit.Workspace -verbose on, off --to-file results_all_2016.txt")
    assert lc.get_main_param() is None and lc.get_string("-folder") == "CARSKit.Workspace"
    assert lc.is_on("-verbose") and lc.params["-verbose"] == ["on", "off"]
    lc = config.LineConfiger("on -topN 10")
    assert lc.is_main_on() and lc.get_int("-topN") == 10
    lc = config.LineConfiger("-lw 0.01 -lf 0.02")
    assert lc.get_float("-lw") == float(np.float32(0.01))


def test_file_configer_reads_setting_conf(tmp_path):
    p = tmp_path / "s.conf"
    p.write_text("# comment\ndataset.ratings.lins=/x/y/ratings.txt\nrecommender=camf_ci\nnum.factors=10\n"
                 "learn.rate=2e-2 -max -1 -bold-driver\n! other comment\nkey\\ with\\ space = v\\\\w\n")
    cf = config.FileConfiger(str(p))
    assert cf.get_path("dataset.ratings") == "/x/y/ratings.txt"
    assert cf.get_int("num.factors") == 10 and cf.get_int("num.max.iter", 100) == 100
    assert cf.get_param_options("learn.rate").contains("-bold-driver")
    assert cf.props["key with space"] == "v\\w"
    c = recommender.Conf(cf)
    assert c.bold_driver and c.init_lrate == 0.019999999552965164 and c.num_factors == 10


def test_java_random_product_equals_oracle_stream():
    for seed in (1, 42, 20260927):
        a, b = javarand.JavaRandom(seed), oracle_np.JavaRandom(seed)
        assert [a.next_double() for _ in range(50)] == [b.next_double() for _ in range(50)]
    assert javarand.JavaRandom(42).next_double() == 0.7275636800328681


def test_split_folds_follows_the_reference_recipe():
    n, k, seed = 23, 5, 1
    labels, kk = splitter.split_folds(n, k, seed)
    assert kk == 5 and sorted(np.bincount(labels)[1:].tolist()) == [4, 4, 5, 5, 5]
    # recipe restated in the open: draw, label by position, sort by draw, deal back in CRS order
    g = oracle_np.JavaRandom(seed)
    rdm = [g.next_double() for _ in range(n)]
    fold = [int(i / (n / 5.0)) + 1 for i in range(n)]
    want = [f for _, f in sorted(zip(rdm, fold))]
    assert labels.tolist() == want
    labels2, kk2 = splitter.split_folds(3, 5, seed)          # more folds than ratings
    assert kk2 == 3 and sorted(labels2.tolist()) == [1, 2, 3]
    data = util.small_data(n=23, seed=2)
    tr, te = splitter.kth_fold(data.subset(np.arange(23)), labels, 2)
    assert tr.n + te.n == 23 and te.n == int((labels == 2).sum())


def _depaul_conf(tmp_path):
    shutil.copyfile(os.path.join(GOLDEN, "depaul_ratings_compact.csv"), tmp_path / "ratings.txt")
    conf = open(os.path.join(GOLDEN, "depaul_setting.conf")).read().replace("PLACEHOLDER_SET_BY_TEST", str(tmp_path / "ratings.txt"))
    (tmp_path / "setting.conf").write_text(conf)
    return str(tmp_path / "setting.conf")


def test_c1_biasedmf_depaul_via_setting_conf(tmp_path):
    """BASELINE config C1: setting.conf -> compact->binary rewrite -> DataDAO -> 5-fold CV -> BiasedMF k=10 ->
    MAE/RMSE, with the oracle as the engine (plumbing; no GPU)."""
    lines = []
    avg, algos, rate_dao = main.run(_depaul_conf(tmp_path), engine_factory=util.OracleEngine, log=lines.append,
                                    conf_overrides={"num_iters": 30})
    assert (rate_dao.num_users, rate_dao.num_items, rate_dao.num_context_dims) == (97, 79, 3)
    # 5 043 lines in the file, 8 of them exact repeats: the transformer keys its HashMap by the whole line, so the
    # rewritten train.csv -- and hence the reference's own run -- has 5 035 rating lines
    assert rate_dao.num_ratings == 5035 and rate_dao.rating_scale == [1.0, 2.0, 3.0, 4.0, 5.0]
    assert len(algos) == 5 and all(a.algo_name == "BiasedMF" for a in algos)
    assert sum(a.testMatrix.n for a in algos) == rate_dao.nnz
    assert lines[-1].startswith("Final Results by BiasedMF, MAE: ") and ", RMSE: " in lines[-1] and "NAME: " in lines[-1]
    assert 0.8 < avg["RMSE"] < 1.3 and 0.5 < avg["MAE"] < 1.1              # sane for DePaulMovie
    # each fold is exactly a direct oracle run on the same split and init (the classes add nothing numerical)
    a = algos[2]
    st = synth.init_state("BiasedMF", a.trainMatrix, 10, seed=a.conf.init_seed)
    orc = util.c_oracle("BiasedMF", a.trainMatrix, 10, st, a.globalMean)
    losses, lrs, _ = orc.build_model(30, util.LR, bold_driver=True)
    assert a.losses == losses.tolist() and a.lrates == lrs.tolist()
    golden = os.path.join(GOLDEN, "golden_c1_depaul_biasedmf.json")
    rec = {"folds": [{"MAE": x.measures["MAE"], "RMSE": x.measures["RMSE"], "n_test": x.testMatrix.n} for x in algos],
           "avg_MAE": avg["MAE"], "avg_RMSE": avg["RMSE"], "iters": 30}
    if os.environ.get("CARSKIT_WRITE_GOLDEN"):
        json.dump(rec, open(golden, "w"), indent=1)
    want = json.load(open(golden))
    assert rec["folds"] == want["folds"] and rec["avg_RMSE"] == want["avg_RMSE"]


def _ranking_conf(tmp_path, algo="camf_cu", topn=10, extra=""):
    conf = _depaul_conf(tmp_path)
    txt = open(conf).read().replace("recommender=biasedmf", "recommender=" + algo)
    txt = txt.replace("item.ranking=off -topN 10", "item.ranking=on -topN %d%s" % (topn, extra))
    txt = txt.replace("-threshold -1", "-threshold 3")
    open(conf, "w").write(txt)
    return conf


def test_item_ranking_via_setting_conf(tmp_path):
    """item.ranking=on: execute() evaluates with evalRankings (Recommender.java:346) and the driver prints the
    Pre/Rec/AUC/MAP/NDCG/MRR line; golden values minted by the oracle (training + ranking) on DePaulMovie."""
    lines = []
    avg, algos, _ = main.run(_ranking_conf(tmp_path), engine_factory=util.OracleEngine, log=lines.append,
                             conf_overrides={"num_iters": 20})
    assert lines[-1].startswith("Final Results by CAMF_CU, Pre5: ") and ",Pre10: " in lines[-1] and ",MRR5: " in lines[-1]
    assert "PreN" not in lines[-1] and "RMSE" not in lines[-1]
    assert all(a.conf.bin_thold == 3.0 and a.conf.is_ranking for a in algos)
    assert 0.5 < avg["AUC10"] < 1.0 and 0.0 < avg["Pre10"] < 0.5 and avg["Rec10"] > avg["Rec5"] > 0.0
    assert avg["PreN"] == avg["Pre10"] and avg["NDCGN"] == avg["NDCG10"]
    golden = os.path.join(GOLDEN, "golden_depaul_camf_cu_ranking.json")
    rec = {"iters": 20, "folds": [{m: a.measures[m] for m in ("Pre10", "Rec10", "AUC10", "MAP10", "NDCG10", "MRR10")} for a in algos]}
    if os.environ.get("CARSKIT_WRITE_GOLDEN"):
        json.dump(rec, open(golden, "w"), indent=1)
    assert rec == json.load(open(golden))


def test_item_ranking_topn_and_strategy_options(tmp_path):
    conf = _ranking_conf(tmp_path, "biasedmf", topn=3, extra=" -ignore 5")
    open(conf, "a").write("\neval.strategy=uc\n")
    lines = []
    avg, algos, _ = main.run(conf, engine_factory=util.OracleEngine, log=lines.append, conf_overrides={"num_iters": 5})
    c = algos[0].conf
    assert (c.num_recs, c.num_ignore, c.eval_strategy) == (3, 5, "uc")
    assert ", Pre3: " in lines[-1] and ",NDCG3: " in lines[-1] and ",MRR3: " in lines[-1]
    assert avg["Pre10"] <= 0.3 + 1e-12          # a 3-long list has at most 3 hits in the Pre@10 numerator


def test_driver_rejects_unaccelerated_recommenders(tmp_path):
    conf = _depaul_conf(tmp_path)
    txt = open(conf).read().replace("recommender=biasedmf", "recommender=itemknn")
    open(conf, "w").write(txt)
    with pytest.raises(ValueError):
        main.run(conf, engine_factory=util.OracleEngine, log=lambda *a: None)


FRAPPE_ZIP = "/root/reference/context-aware_data_sets/Mobile_Frappe.zip"


@pytest.mark.skipif(not os.path.exists(FRAPPE_ZIP), reason="the Frappe data may not be redistributed; only where the reference is mounted")
def test_frappe_loads_through_transformer_and_dao(tmp_path):
    """BASELINE config C2's data path on the real file (tab-separated -> comma CSV first, as CARSKit requires)."""
    import zipfile
    from carskit_amd import dao
    from oracle import dao_oracle
    raw = zipfile.ZipFile(FRAPPE_ZIP).read("Mobile_Frappe/frappe/frappe.csv").decode("utf-8")
    src = tmp_path / "frappe_compact.csv"
    src.write_text("\n".join(",".join(line.split("\t")) for line in raw.splitlines()) + "\n")
    out = tmp_path / "train.csv"
    tree = dao.transform_compact_to_binary(str(src), str(out))
    d = dao.DataDAO(str(out))
    assert (d.num_users, d.num_items, d.num_context_dims) == (957, 4082, 8)
    assert d.num_ratings == 96203 and d.num_conditions == 7 + 7 + 2 + 3 + 2 + 9 + 80 + 233
    want, max_bin = dao_oracle.compact_to_binary(str(src))
    got = open(out).read().split("\n")[:-1]
    assert got[0] == want[0] and sorted(got[1:]) == sorted(want[1:])
    if max_bin < 8:              # no bin could have been treeified: the row ORDER is the reference's too
        assert got == want and not tree


def test_test_set_evaluation_via_setting_conf(tmp_path):
    """`evaluation.setup=test-set -f <file>`: both files transformed against the merged conditions, the test DAO
    shares and extends the training DAO's ids, the recommender is sized for the union."""
    shutil.copyfile(os.path.join(GOLDEN, "train_compact.csv"), tmp_path / "ratings.txt")
    shutil.copyfile(os.path.join(GOLDEN, "test_loose.csv"), tmp_path / "test.txt")
    conf = open(os.path.join(GOLDEN, "depaul_setting.conf")).read().replace("PLACEHOLDER_SET_BY_TEST", str(tmp_path / "ratings.txt"))
    conf = conf.replace("evaluation.setup=cv -k 5 -p off --rand-seed 1 --test-view all",
                        "evaluation.setup=test-set -f %s --rand-seed 1" % (tmp_path / "test.txt"))
    conf = conf.replace("recommender=biasedmf", "recommender=camf_cu")
    (tmp_path / "setting.conf").write_text(conf)
    lines = []
    avg, algos, rate_dao = main.run(str(tmp_path / "setting.conf"), engine_factory=util.OracleEngine, log=lines.append,
                                    conf_overrides={"num_iters": 5})
    a = algos[0]
    assert a.algo_name == "CAMF_CU" and a.testMatrix.n > 0
    assert a.trainMatrix.n_users >= rate_dao.num_users              # union of train and test users
    assert a.state["P"].shape == (a.trainMatrix.n_users, 10)
    assert lines[-1].startswith("Final Results by CAMF_CU, MAE: ")
    assert np.isfinite(avg["RMSE"])


EXE = os.path.join(os.path.dirname(GOLDEN), "..", "carskit_amd", "bin", "carskit-mi355x")


def _cpp_init(model, train, k, seed):
    """The C++ host's initModel(): java.util.Random(seed) -> P, Q ~ N(0,0.1) row-major, then the biases in source order
    (carskit_amd/csrc/host/recommender.hpp), reproduced with the oracle's independent JRandom."""
    from oracle import oracle_c
    g = oracle_c.JRandom(seed)
    st = {"P": g.gaussian((train.n_users, k)), "Q": g.gaussian((train.n_items, k))}
    if model in ("BiasedMF", "CAMF_C"):
        st["userBias"], st["itemBias"] = g.gaussian(train.n_users), g.gaussian(train.n_items)
        if model == "CAMF_C":
            st["condBias"] = g.gaussian(train.n_conds)
    elif model == "CAMF_CI":
        st["userBias"], st["icBias"] = g.gaussian(train.n_users), g.uniform((train.n_items, train.n_conds))
    elif model == "CAMF_CU":
        st["itemBias"], st["ucBias"] = g.gaussian(train.n_items), g.uniform((train.n_users, train.n_conds))
    elif model == "CAMF_CUCI":
        st["ucBias"], st["icBias"] = g.gaussian((train.n_users, train.n_conds)), g.gaussian((train.n_items, train.n_conds))
    elif model == "SVD++":
        st["userBias"], st["itemBias"] = g.gaussian(train.n_users), g.gaussian(train.n_items)
        st["Y"] = g.gaussian((train.n_items, k))
    elif model == "CAMF_ICS":     # P, Q re-drawn uniform on top of the gaussian draws (CAMF_ICS.java:36-51)
        st["P"], st["Q"] = g.uniform((train.n_users, k)), g.uniform((train.n_items, k))
        st["ccMatrix"] = np.ones((train.n_conds, train.n_conds))
    elif model == "CAMF_LCS":
        st["cfMatrix"] = g.uniform((train.n_conds, int(train.meta.get("num_f", 10))))
    elif model == "CAMF_MCS":
        st["cVector"] = g.uniform(train.n_conds) * (1.0 / np.sqrt(max(1, train.n_dims)))
    return st


def expected_from_oracle(conf_path, model_cls_name, iters, engine_factory=None):
    """What the C++ driver must print for a cv run: same transformer/DAO/fold assignment (shared C ABI + the Python
    splitter, itself checked against the recipe), the C++ init stream, the oracle as the engine (or `engine_factory`)."""
    class Cls(recommender.RECOMMENDERS[model_cls_name]):
        def initModel(self):
            self.trainMatrix.meta["num_f"] = self.conf.num_f
            self.state = _cpp_init(self.algo_name, self.trainMatrix, self.numFactors, self.conf.init_seed)
    saved = recommender.RECOMMENDERS[model_cls_name]
    recommender.RECOMMENDERS[model_cls_name] = Cls
    try:
        avg, _, _ = main.run(conf_path, engine_factory=engine_factory or util.OracleEngine, log=lambda *a: None,
                             conf_overrides={"num_iters": iters})
    finally:
        recommender.RECOMMENDERS[model_cls_name] = saved
    return avg


def test_cpp_host_driver_loads_data_then_fails_loudly_without_gpu(tmp_path):
    import subprocess
    from carskit_amd import capi
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    conf = _depaul_conf(tmp_path)
    p = subprocess.run([EXE, "-c", conf, "--iters", "3"], capture_output=True, text=True)
    assert "Rating data set has been successfully loaded." in p.stdout and "With Setup: cv -k 5" in p.stdout
    assert p.returncode == 1 and "no HIP device" in p.stderr and "no CPU fallback" in p.stderr
    q = subprocess.run([EXE, "-c", str(tmp_path / "missing.conf")], capture_output=True, text=True)
    assert q.returncode == 1 and "cannot open configuration file" in q.stderr


@pytest.mark.parametrize("algo,name", [("svd++", "SVD++"), ("camf_ics", "CAMF_ICS"), ("camf_lcs -f 6", "CAMF_LCS"), ("camf_mcs", "CAMF_MCS")])
def test_n1_recommenders_via_setting_conf(tmp_path, algo, name):
    """`recommender=svd++|camf_ics|camf_lcs|camf_mcs` resolve (CARSKit.java:469,708-712), the transformer's ':na'
    conditions reach the models as EmptyContextConditions, the three similarity models evaluate as top-N recommenders whatever
    item.ranking says, and every fold equals a direct oracle run (plumbing; the oracle is the engine here, no GPU)."""
    conf = _depaul_conf(tmp_path)
    txt = open(conf).read().replace("recommender=biasedmf", "recommender=" + algo).replace("learn.rate=2e-2", "learn.rate=2e-3")
    open(conf, "w").write(txt)
    lines = []
    avg, algos, rate_dao = main.run(conf, engine_factory=util.OracleEngine, log=lines.append, conf_overrides={"num_iters": 6})
    assert len(algos) == 5 and all(a.algo_name == name for a in algos)
    assert rate_dao.empty_context_conditions and len(rate_dao.empty_context_conditions) == rate_dao.num_context_dims
    a = algos[1]
    assert all(np.isfinite(a.losses)) and a.losses[-1] < a.losses[0]
    if name == "SVD++":
        assert lines[-1].startswith("Final Results by SVD++, MAE: ") and 0.5 < avg["RMSE"] < 2.0
    else:
        assert a.conf.is_ranking and lines[-1].startswith("Final Results by %s, Pre5: " % name) and 0.4 < avg["AUC10"] <= 1.0
        assert a.conf.num_f == (6 if name == "CAMF_LCS" else 10)
        if name == "CAMF_LCS":
            assert a.state["cfMatrix"].shape == (rate_dao.num_conditions, 6)
    # the fold is exactly a direct oracle run on the same split and init
    from oracle import oracle_c
    a.trainMatrix.meta["num_f"] = a.conf.num_f
    st = synth.init_state(name, a.trainMatrix, 10, seed=a.conf.init_seed)
    u, j, ctx, r = a.train_tuples()
    orc = oracle_c.SimOracle(name, 10, a.numUsers, a.numItems, a.numConditions, u, j, ctx, r, a.trainMatrix.ctx_ptr, a.trainMatrix.ctx_conds,
                             a.trainMatrix.empty_conds, st, a.globalMean, a.conf.regU, a.conf.regI, a.conf.regB, a.conf.regC,
                             n_ctx_dims=a.trainMatrix.n_dims)
    lr, last, losses = a.conf.init_lrate, 0.0, []
    for it in range(1, 7):
        losses.append(orc.epoch(lr))
        if it > 1:
            lr = lr * 1.05 if abs(last) > abs(losses[-1]) else lr * 0.5
        last = losses[-1]
    assert a.losses == losses

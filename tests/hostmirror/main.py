"""Host-flow MIRROR, test plumbing only (it lives under tests/): the product host is the C++ one (carskit_amd/csrc/host, built to
carskit_amd/bin/carskit-mi355x).  run() below restates the same flow in
Python for one reason: it takes an `engine_factory`, so the tests can put the CPU ORACLE behind the identical host logic (config parsing, splits,
bold driver, early stop, measures) and compare the two hosts and the two engines line by line (tests/test_host_layer.py,
tests/test_gpu_realdata.py).  It is test plumbing, not a second product.

The reference driver's flow for the accelerated recommenders
(src/carskit/main/CARSKit.java: execute :109, preset :140, readData :220, runAlgorithm :310, runCrossValidation :388):
load the config, bring the rating file to the binary format (DataTransformer), read it (DataDAO), split
(`cv -k N` follows the reference's seeded fold assignment; `test-set`; `given-ratio` uses a seeded draw because the
reference's Math.random() is unseedable), run the recommender per fold, average the measures, print
`Final Results by <algo>, MAE: ..., RMSE: ...` (or the Pre/Rec/AUC/MAP/NDCG/MRR line when item.ranking=on)."""
import argparse
import os
import sys

import numpy as np

from carskit_amd import dao, synth

from . import splitter
from .config import FileConfiger, LineConfiger
from .recommender import RECOMMENDERS, Conf, get_eval_info


def read_data(cf, log):
    """CARSKit.readData (:220-273): transform the rating file(s) into <workspace>/train.csv (and test.csv for
    `test-set -f <file>`), then read them with the DataDAO.  Returns (train dao, test dao or None, workspace)."""
    rating_file = cf.get_path("dataset.ratings")
    if rating_file is None or not os.path.exists(rating_file):
        raise FileNotFoundError("Your rating file path is incorrect: File doesn't exist. Please double check your configuration.")
    out = cf.get_param_options("output.setup")
    folder = out.get_string("-folder", "CARSKit.Workspace") if out else "CARSKit.Workspace"
    work = os.path.join(os.path.dirname(os.path.abspath(rating_file)), folder) + os.sep
    os.makedirs(work, exist_ok=True)
    log("WorkingPath: " + work)
    ev = LineConfiger(cf.get_string("evaluation.setup"))
    test_file = ev.get_string("-f") if (ev.get_main_param() or "").lower().strip() == "test-set" else None
    ro = cf.get_param_options("ratings.setup")
    if ro is None or ro.get_int("-datatransformation", 1) > 0:
        fmt = dao.validate_data_format(rating_file)
        if fmt in (2, 3):
            log("You rating data is in %s format. CARSKit is working on transformation on the data format..."
                % ("Loose" if fmt == 2 else "Compact"))
        dao.transform(rating_file, work + "train.csv", test_file, work + "test.csv" if test_file else None)
    train = dao.DataDAO(work + "train.csv")
    test = dao.DataDAO(work + "test.csv", train=train) if test_file else None
    log("Rating data set has been successfully loaded.")
    return train, test, work


def run(config_path, engine_factory=None, log=print, conf_overrides=None):
    cf = FileConfiger(config_path)
    rate_dao, test_dao, work = read_data(cf, log)
    data = rate_dao.rating_data()
    conf = Conf(cf, **(conf_overrides or {}))
    algo_line = LineConfiger(cf.get_string("recommender"))
    name = algo_line.get_main_param().lower()
    if name not in RECOMMENDERS:
        raise ValueError("recommender '%s' is not on the accelerated path (supported: %s)" % (name, ", ".join(RECOMMENDERS)))
    cls = RECOMMENDERS[name]
    setup = cf.get_string("evaluation.setup")
    ev = LineConfiger(setup)
    log("With Setup: " + setup)
    seed = ev.get_long("--rand-seed", 1)
    mode = (ev.get_main_param() or "").lower()
    algos = []
    if mode == "cv":
        k = ev.get_int("-k", 5)
        labels, k = splitter.split_folds(data.n, k, seed)
        from carskit_amd import capi
        ngpu = capi.device_count() if engine_factory is None else 0
        for f in range(1, k + 1):
            train, test = splitter.kth_fold(data, labels, f)
            algo = cls(train, test, f, conf, engine_factory, log)
            if ngpu > 1:
                algo.device = (f - 1) % ngpu                          # fold -> GPU round robin
            algos.append(algo)
        if ev.is_on("-p", True) and engine_factory is None and len(algos) > 1:
            # `cv -p on`: one thread per fold (CARSKit.java:395-412); every fold owns its handle and stream, and the
            # library calls release the GIL, so the folds' epochs overlap on the GPU(s)
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=len(algos)) as pool:
                list(pool.map(lambda a: a.execute(), algos))
        else:
            for algo in algos:
                algo.execute()
    elif mode == "test-set":
        # the id spaces are the union's: test-only users/items exist in the model (with their initial values) exactly
        # as in the reference, where rateDao.numUsers() is read after the test DAO extended the shared maps
        test = test_dao.rating_data()
        train = synth.RatingData(test.n_users, test.n_items, data.n_conds, data.n_dims, data.u, data.j, data.ctx, data.r,
                                 test.ctx_ptr, test.ctx_conds, data.min_rate, data.max_rate, dict(data.meta), data.empty_conds)
        test.min_rate, test.max_rate = data.min_rate, data.max_rate      # rating scale of the TRAINING dao (:198-200)
        algo = cls(train, test, -1, conf, engine_factory, log)
        algo.execute()
        algos.append(algo)
    else:
        ratio = ev.get_double("-r", 0.8)
        train, test = synth.split(data, 1.0 - ratio, seed=seed)
        algo = cls(train, test, -1, conf, engine_factory, log)
        algo.execute()
        algos.append(algo)
    avg = {}
    for a in algos:
        for m, v in a.measures.items():
            avg[m] = avg.get(m, 0.0) + v / len(algos)
    info = "Final Results by %s, %s" % (algos[0].algo_name, get_eval_info(avg, algos[0].conf))   # (top-N models force ranking)
    log(info)
    return avg, algos, rate_dao

"""`python -m carskit_amd.main -c setting.conf` -- the reference driver's flow for the accelerated recommenders
(src/carskit/main/CARSKit.java: execute :109, preset :140, readData :220, runAlgorithm :310, runCrossValidation :388):
load the config, bring the rating file to the binary format (DataTransformer), read it (DataDAO), split
(`cv -k N` follows the reference's seeded fold assignment; `test-set`; `given-ratio` uses a seeded draw because the
reference's Math.random() is unseedable), run the recommender per fold, average the measures, print
`Final Results by <algo>, MAE: ..., RMSE: ...`.  Only rating prediction (item.ranking=off semantics) is covered."""
import argparse
import os
import shutil
import sys

import numpy as np

from . import dao, splitter, synth
from .config import FileConfiger, LineConfiger
from .recommender import RECOMMENDERS, Conf, get_eval_info


def validate_data_format(path):
    """CARSKit.validateDataFormat (:179-215): 1 binary, 2 loose, 3 compact."""
    with open(path, encoding="latin-1") as f:
        header = f.readline().rstrip("\r\n").split(",")
        data = f.readline().rstrip("\r\n").split(",")
    if len(header) >= 2 and header[-2].strip().lower() == "dimension" and header[-1].strip().lower() == "condition":
        return 2
    for i in range(3, len(header)):
        tok = data[i] if i < len(data) else ""
        if ":" not in header[i] or not (tok.strip().lstrip("-").isdigit() and set(tok.strip().lstrip("-")) <= set("01")):
            return 3
    return 1


def read_data(cf, log):
    rating_file = cf.get_path("dataset.ratings")
    if rating_file is None or not os.path.exists(rating_file):
        raise FileNotFoundError("Your rating file path is incorrect: File doesn't exist. Please double check your configuration.")
    out = cf.get_param_options("output.setup")
    folder = out.get_string("-folder", "CARSKit.Workspace") if out else "CARSKit.Workspace"
    work = os.path.join(os.path.dirname(os.path.abspath(rating_file)), folder) + os.sep
    os.makedirs(work, exist_ok=True)
    log("WorkingPath: " + work)
    train_csv = work + "train.csv"
    fmt = validate_data_format(rating_file)
    if fmt == 1:
        shutil.copyfile(rating_file, train_csv)
    elif fmt == 3:
        log("You rating data is in Compact format. CARSKit is working on transformation on the data format...")
        dao.transform_compact_to_binary(rating_file, train_csv)
    else:
        raise NotImplementedError("loose-format input: convert to the compact or binary format first")
    d = dao.DataDAO(train_csv)
    log("Rating data set has been successfully loaded.")
    return d, work


def run(config_path, engine_factory=None, log=print, conf_overrides=None):
    cf = FileConfiger(config_path)
    rate_dao, work = read_data(cf, log)
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
        for f in range(1, k + 1):
            train, test = splitter.kth_fold(data, labels, f)
            algo = cls(train, test, f, conf, engine_factory, log)
            algo.execute()
            algos.append(algo)
    elif mode == "test-set":
        raise NotImplementedError("test-set evaluation needs the shared-id test DAO (next row N2)")
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
    info = "Final Results by %s, %s" % (algos[0].algo_name, get_eval_info(avg))
    log(info)
    return avg, algos, rate_dao


def main(argv=None):
    ap = argparse.ArgumentParser(prog="carskit_amd")
    ap.add_argument("-c", dest="configs", action="append", default=None)
    args = ap.parse_args(argv)
    for c in args.configs or ["setting.conf"]:
        run(c)


if __name__ == "__main__":
    main(sys.argv[1:])

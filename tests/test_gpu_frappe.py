"""BASELINE.json configs[1] on the REAL Frappe file: "CAMF_C k=64 fp32 on Frappe, 1xMI355X (first HIP kernel, parity vs Java)".
tests/golden/frappe_compact.csv.gz -> product cmi_validate_data_format -> cmi_transform -> cmi_dao_read (all array-equal with
oracle/dao_oracle.py: tests/test_frappe_data.py) -> 5-fold CV -> CAMF_C k=64, 15 bold-driver epochs per fold on the GPU through the
C ABI, against the oracle behind the same host logic on the same folds and init:
  * fp32 state (the config's dtype): identical learning-rate trajectory (= identical bold-driver decisions), loss to 2e-5 relative,
    RMSE / MAE within 1e-5 (north_star's bar) on every fold;
  * fp64 + strict: losses and every model array bit-identical;
  * the raw usage counts at the reference's default rate overflow in epoch 1 on both sides (same fatal error), and train bit-identically
    at a rate that keeps them finite;
  * the C++ host (carskit-mi355x -c setting.conf) prints the oracle's numbers.
Reference: CAMF_C.java:75-138, IterativeRecommender.java:145-229, CARSKit.java:179-215,388-412."""
import re
import subprocess

import numpy as np
import pytest

from carskit_amd import capi
from tests import frappe, util
from tests.hostmirror import main, recommender

pytestmark = pytest.mark.gpu
STRICT = capi.FLAG_STATE_F64 | capi.FLAG_STRICT | capi.FLAG_SCHED_SERIAL


def test_c2_camf_c_k64_fp32_on_frappe(tmp_path):
    conf = frappe.write_conf(tmp_path, "log")
    lines = []
    avg_g, gpu, rate_dao = main.run(conf, log=lines.append)
    avg_c, cpu, _ = main.run(conf, engine_factory=util.OracleEngine, log=lambda *a: None)
    assert (rate_dao.num_users, rate_dao.num_items, rate_dao.num_conditions, rate_dao.nnz) == (957, 4082, 343, 96195)
    assert len(gpu) == 5 and lines[-1].startswith("Final Results by CAMF_C, MAE: ")
    for a, b in zip(gpu, cpu):
        assert a.numFactors == 64 and len(a.losses) == 15
        assert a.lrates == b.lrates                                            # same bold-driver decisions, epoch by epoch
        np.testing.assert_allclose(a.losses, b.losses, rtol=2e-5)
        assert abs(a.measures["RMSE"] - b.measures["RMSE"]) <= 1e-5 and abs(a.measures["MAE"] - b.measures["MAE"]) <= 1e-5
        assert a.engine.inst.schedule_info()["kind"] == "serial"
    assert abs(avg_g["RMSE"] - avg_c["RMSE"]) <= 1e-5 and abs(avg_g["MAE"] - avg_c["MAE"]) <= 1e-5


def test_c2_frappe_strict_fp64_bit_exact(tmp_path):
    # (the strict serial wave walks one tuple at a time in the reference's summation order, ~8 us per tuple at k = 64: two folds x
    # four epochs keep this test at a few seconds; the fp32 test above runs the full 5 x 15)
    conf = frappe.write_conf(tmp_path, "log", folds=2)
    _, gpu, _ = main.run(conf, log=lambda *a: None, conf_overrides={"flags": STRICT, "num_iters": 4})
    _, cpu, _ = main.run(conf, engine_factory=util.OracleEngine, log=lambda *a: None, conf_overrides={"num_iters": 4})
    assert len(gpu) == 2
    for a, b in zip(gpu, cpu):
        assert a.losses == b.losses and a.lrates == b.lrates
        for name, arr in a.state.items():
            assert np.array_equal(arr, b.state[name].reshape(arr.shape)), name
        # the model is bit-identical; evalRatings sums |r - pred| over the test tuples in a tree on the device: 1e-12
        assert abs(a.measures["RMSE"] - b.measures["RMSE"]) <= 1e-12 and abs(a.measures["MAE"] - b.measures["MAE"]) <= 1e-12


def test_c2_frappe_raw_counts(tmp_path):
    conf = frappe.write_conf(tmp_path, "raw")
    with pytest.raises(FloatingPointError, match="Loss = NaN or Infinity"):     # IterativeRecommender.java:181-184
        main.run(conf, log=lambda *a: None)
    conf = frappe.write_conf(tmp_path, "raw", folds=2)
    over = {"init_lrate": 1e-6, "flags": STRICT, "num_iters": 4}
    _, gpu, _ = main.run(conf, log=lambda *a: None, conf_overrides=over)
    _, cpu, _ = main.run(conf, engine_factory=util.OracleEngine, log=lambda *a: None, conf_overrides={"init_lrate": 1e-6, "num_iters": 4})
    for a, b in zip(gpu, cpu):
        assert a.losses == b.losses and a.lrates == b.lrates and np.isfinite(a.losses).all()
        for name, arr in a.state.items():
            assert np.array_equal(arr, b.state[name].reshape(arr.shape)), name


def test_c2_frappe_through_the_cpp_host(tmp_path):
    from tests.test_host_layer import EXE, expected_from_oracle
    conf = frappe.write_conf(tmp_path, "log")
    strict_conf = frappe.write_conf(tmp_path / "strict", "log", folds=2)
    want2 = expected_from_oracle(strict_conf, "camf_c", 4)
    p = subprocess.run([EXE, "-c", strict_conf, "--iters", "4", "--flags", str(STRICT), "--precise"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    m = re.search(r"PRECISE CAMF_C folds=2 MAE=(\S+) RMSE=(\S+)", p.stdout)
    assert m, p.stdout[-500:]
    assert abs(float(m.group(1)) - want2["MAE"]) <= 1e-12 and abs(float(m.group(2)) - want2["RMSE"]) <= 1e-12
    want = expected_from_oracle(conf, "camf_c", 15)
    p32 = subprocess.run([EXE, "-c", conf, "--precise"], capture_output=True, text=True)      # the config's dtype: fp32 state, 5 folds x 15 epochs
    m32 = re.search(r"PRECISE CAMF_C folds=5 MAE=(\S+) RMSE=(\S+)", p32.stdout)
    assert m32, p32.stdout[-500:] + p32.stderr
    assert abs(float(m32.group(1)) - want["MAE"]) <= 1e-5 and abs(float(m32.group(2)) - want["RMSE"]) <= 1e-5
    # raw counts at the default rate: the reference's fatal error, non-zero exit
    raw = frappe.write_conf(tmp_path, "raw")
    q = subprocess.run([EXE, "-c", raw], capture_output=True, text=True)
    assert q.returncode != 0 and "Loss = NaN or Infinity" in (q.stderr + q.stdout)

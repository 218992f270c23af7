"""BASELINE configs[1]'s data set, the real Frappe file, through the product's data path on the CPU:
CARSKit.validateDataFormat (CARSKit.java:179-215) -> DataTransformer compact->binary (DataTransformer.java:231-329) ->
DataDAO.readData (DataDAO.java:196-354), every id table and matrix cell array-equal with the independent restatement in
oracle/dao_oracle.py on the same file.  The training parity on this file is tests/test_gpu_frappe.py."""
import numpy as np
import pytest

from carskit_amd import dao
from oracle import dao_oracle
from tests import frappe, util
from tests.hostmirror import main


def test_frappe_validate_transform_read_equal_the_restatement(tmp_path):
    src = frappe.write_ratings(tmp_path, "raw")
    assert dao.validate_data_format(src) == 3                         # compact
    out = str(tmp_path / "train.csv")
    assert dao.transform(src, out) is False                          # no HashMap bin treeified: the row order is the reference's
    want_lines, _, max_bin = dao_oracle.transform(src)
    assert max_bin < 8
    got = open(out, encoding="latin-1").read().split("\n")
    assert got[-1] == "" and got[:-1] == want_lines
    d = dao.DataDAO(out)
    assert (d.num_users, d.num_items, d.num_conditions, d.num_context_dims) == (957, 4082, 343, 8)
    # 96 203 lines, none repeated verbatim; 8 (user,item,context) cells appear twice with different counts: the last one wins in
    # the Table (DataDAO.java:342) while numRatings counts every line (:234,348)
    assert d.num_ratings == 96203 and d.nnz == 96195 and d.empty_context_conditions == []
    assert d.rating_scale[0] == 1.0 and d.rating_scale[-1] == 28752.0 and len(d.rating_scale) == 1981
    w = dao_oracle.read_data(out)
    assert d.raw_ids("user") == w["users"] and d.raw_ids("item") == w["items"]
    assert d.raw_ids("ui") == w["uis"] and d.raw_ids("ctx") == w["ctxs"]
    assert d.ui_user.tolist() == w["ui_user"] and d.ui_item.tolist() == w["ui_item"]
    assert [d.ctx_conds[d.ctx_ptr[c]:d.ctx_ptr[c + 1]].tolist() for c in range(d.num_contexts)] == w["ctx_conds"]
    assert d.ui.tolist() == w["ui"] and d.ctx.tolist() == w["ctx"] and d.r.tolist() == w["r"]
    assert all(len(c) == 8 for c in w["ctx_conds"])                    # every rating has all 8 dimensions: D = 8


def test_frappe_raw_counts_overflow_at_the_default_rate_like_the_reference(tmp_path):
    """Raw usage counts (up to 28 752) with learn.rate=2e-2: the first epoch's loss is not finite, and the host reports the reference's
    fatal error (IterativeRecommender.java:181-184) -- plumbing, with the oracle as the engine."""
    conf = frappe.write_conf(tmp_path, "raw")
    with pytest.raises(FloatingPointError, match="Loss = NaN or Infinity"):
        main.run(conf, engine_factory=util.OracleEngine, log=lambda *a: None)


def test_frappe_log_scale_trains_with_the_oracle(tmp_path):
    conf = frappe.write_conf(tmp_path, "log")
    avg, algos, rate_dao = main.run(conf, engine_factory=util.OracleEngine, log=lambda *a: None)
    assert len(algos) == 5 and all(a.algo_name == "CAMF_C" and len(a.losses) == 15 for a in algos)
    assert sum(a.testMatrix.n for a in algos) == rate_dao.nnz == 96195
    assert all(np.isfinite(a.losses).all() and a.losses[-1] < a.losses[0] for a in algos)
    assert 0.5 < avg["RMSE"] < 1.0

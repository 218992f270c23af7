"""the id-mapper against the reference's OWN `DataDAO.readData`, interpreted from its Java source by
oracle/jvm/javasrc.py (oracle/mint_reference_dao.py -> tests/golden/reference_dao.json).  The product's C++ DataDAO (through the C ABI,
no GPU involved) and the Python restatement oracle/dao_oracle.py must reproduce every id table, the CRS order of the rating cells, the
condition lists, EmptyContextConditions and ratingScale exactly -- integers, strings and doubles (as hex)."""
import json
import os

import pytest

from carskit_amd import dao
from oracle import dao_oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(GOLDEN, "reference_dao.json")))["cases"]


def _path(tmp_path, case, key_file, key_text, name):
    if key_file in case:
        return os.path.join(GOLDEN, case[key_file])
    p = tmp_path / name
    with open(p, "w", newline="") as fh:
        fh.write(case[key_text])
    return str(p)


def _check_product(d, e):
    assert (d.num_users, d.num_items, d.num_user_items, d.num_contexts, d.num_conditions, d.num_context_dims) == tuple(e["counts"])
    assert d.raw_ids("user") == e["users"] and d.raw_ids("item") == e["items"]
    assert d.raw_ids("ui") == e["uis"] and d.raw_ids("ctx") == e["ctxs"]
    assert d.raw_ids("cond") == e["conds"] and d.raw_ids("dim") == e["dims"]
    assert d.cond_dim.tolist() == e["cond_dim"] and d.empty_context_conditions == e["empty"]
    assert d.ui_user.tolist() == e["ui_user"] and d.ui_item.tolist() == e["ui_item"]
    assert [d.ctx_conds[d.ctx_ptr[c]:d.ctx_ptr[c + 1]].tolist() for c in range(d.num_contexts)] == e["ctx_conds"]
    assert d.num_ratings == e["num_ratings"]
    assert [float(x).hex() for x in d.rating_scale] == e["scale"]
    assert [[int(a), int(b), float(c).hex()] for a, b, c in zip(d.ui, d.ctx, d.r)] == e["cells"]


def _check_oracle(o, e, shared=False):
    assert o["users"] == e["users"] and o["items"] == e["items"] and o["uis"] == e["uis"] and o["ctxs"] == e["ctxs"]
    assert o["ui_user"] == e["ui_user"] and o["ui_item"] == e["ui_item"] and o["num_ratings"] == e["num_ratings"]
    assert [[a, b, float(c).hex()] for a, b, c in zip(o["ui"], o["ctx"], o["r"])] == e["cells"]
    if not shared:
        assert o["conds"] == e["conds"] and o["dims"] == e["dims"] and o["cond_dim"] == e["cond_dim"] and o["empty"] == e["empty"]
        assert o["ctx_conds"] == e["ctx_conds"] and [float(x).hex() for x in o["scale"]] == e["scale"]


@pytest.mark.parametrize("case", [c for c in CASES if "expect" in c], ids=lambda c: c["name"])
def test_readdata_matches_the_interpreted_reference(tmp_path, case):
    e = case["expect"]
    assert e["statements"] > 500          # the golden came from executed source, not from a table typed in
    path = _path(tmp_path, case, "file", "text", "r.csv")
    if "after_text" in case or "after_file" in case:
        first = _path(tmp_path, case, "after_file", "after_text", "first.csv")
        train = dao.DataDAO(first)
        _check_product(dao.DataDAO(path, train=train), e)
        _check_oracle(dao_oracle.read_data_shared(first, path), e, shared=True)
    else:
        _check_product(dao.DataDAO(path), e)
        _check_oracle(dao_oracle.read_data(path), e)


@pytest.mark.parametrize("case", [c for c in CASES if "throws" in c], ids=lambda c: c["name"])
def test_shared_read_with_another_header_throws_like_the_reference(case):
    """HashBiMap.put refuses a value bound to another key: reading the sample test file over the sample training file's maps throws in
    the reference (its header lists other condition columns); the product refuses the pair too."""
    assert "IllegalArgumentException" in case["throws"]
    train = dao.DataDAO(os.path.join(GOLDEN, case["after_file"]))
    with pytest.raises(Exception):
        dao.DataDAO(os.path.join(GOLDEN, case["file"]), train=train)


# ---- DataTransformer.run(), interpreted from the reference's source (tests/golden/reference_transform.json)
TCASES = json.load(open(os.path.join(GOLDEN, "reference_transform.json")))["cases"]


@pytest.mark.parametrize("case", TCASES, ids=lambda c: c["name"])
def test_transform_matches_the_interpreted_reference(tmp_path, case):
    e = case["expect"]
    a = _path(tmp_path, case, "train_file", "train_text", "in_train.csv")
    b = _path(tmp_path, case, "test_file", "test_text", "in_test.csv") if ("test_file" in case or "test_text" in case) else None
    assert dao.validate_data_format(a) == case["train_format"] and (b is None or dao.validate_data_format(b) == case["test_format"])
    oa, ob = str(tmp_path / "train.csv"), str(tmp_path / "test.csv")
    tree = dao.transform(a, oa, b, ob if b else None)
    assert not tree
    assert open(oa, newline="").read() == e["train.csv"]
    if b:
        assert open(ob, newline="").read() == e["test.csv"]
    else:
        assert "test.csv" not in e
    wa, wb, max_bin = dao_oracle.transform(a, b)
    assert max_bin < 8
    assert "".join(x + "\n" for x in wa) == e["train.csv"] and (b is None or "".join(x + "\n" for x in wb) == e["test.csv"])
    if case["train_format"] != 1 or b:
        assert e["statements"] > 1000


VCASES = json.load(open(os.path.join(GOLDEN, "reference_transform.json")))["validate"]


@pytest.mark.parametrize("case", VCASES, ids=lambda c: c["name"])
def test_validate_data_format_matches_the_interpreted_reference(tmp_path, case):
    """CARSKit.validateDataFormat from source: the format flag, and 0 from the product exactly where the reference throws."""
    path = _path(tmp_path, case, "file", "text", "f.csv")
    want = case["expect"].get("format", 0)
    assert ("throws" in case["expect"]) == (want == 0)
    assert dao.validate_data_format(path) == want
    assert dao_oracle.validate_format(dao_oracle.read_lines(path)) == want


# ---- DataSplitter.splitFolds / getKthFold, interpreted from source over happy.coding's Randoms + Sortor bytecode
SCASES = json.load(open(os.path.join(GOLDEN, "reference_transform.json")))["splitter"]
DRIVER = os.path.join(os.path.dirname(GOLDEN), "..", "carskit_amd", "bin", "carskit-mi355x")


@pytest.mark.parametrize("case", SCASES, ids=lambda c: c["name"])
def test_fold_assignment_matches_the_interpreted_reference(case):
    """The fold label of every matrix entry (CRS order) and the train / test cells of every fold: the C++ driver's split_folds
    (`--print-folds`, no GPU involved) and the host mirror's, against DataSplitter.java executed."""
    import subprocess
    from tests.hostmirror import splitter
    n = len(case["cells"])
    assert case["statements"] > 50
    out = subprocess.run([DRIVER, "--print-folds", str(n), str(case["kfold"]), str(case["seed"])], capture_output=True, text=True, check=True)
    got = [int(x) for x in out.stdout.split()]
    assert got[0] == case["num_fold"] and got[1:] == case["labels"]
    labels, nf = splitter.split_folds(n, case["kfold"], case["seed"])
    assert nf == case["num_fold"] and labels.tolist() == case["labels"]
    # getKthFold: fold f's entries are the TEST matrix, the rest the training matrix, both in CRS order
    for f, fold in enumerate(case["folds"], start=1):
        test = [[r, c, float(v).hex()] for (r, c, v), lab in zip(case["cells"], case["labels"]) if lab == f]
        train = [[r, c, float(v).hex()] for (r, c, v), lab in zip(case["cells"], case["labels"]) if lab != f]
        assert fold["test"] == test and fold["train"] == train

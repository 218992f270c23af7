"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/carskit_mi355x.h
declares, refuses to compute without a GPU (no CPU fallback), and the host-only level scheduler
produces a valid, minimal, order-preserving schedule."""
import os
import re

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from carskit_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "carskit_mi355x.h")).read()
    declared = set(re.findall(r"\b(cmi_[a-z_0-9]+)\s*\(", hdr))
    bound = {name for name, _, _ in capi.SYMBOLS}
    assert declared == bound, declared ^ bound
    L = capi.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.cmi_abi_version() == 5


def test_no_cpu_fallback():
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(capi.CmiError) as ei:
        capi.Instance("CAMF_CI", 8, 10, 10, 4)
    assert ei.value.code == capi.E_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_product_package_never_imports_oracle():
    """No file of the product package -- nor of tools/, include/, java/, jni/ -- imports, loads or links anything under oracle/
    (comments may mention it).  Only tests/ (incl. tests/tools/), __graft_entry__.smoke() and bench.py's cpu_baseline leg do."""
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b|from\s+\.\.?oracle\b)|libcarskit_oracle|oracle_c\b|oracle_np\b|"
                     r"carskit_oracle\.h|[\"'/]oracle/", re.M)
    for top in ("carskit_amd", "tools", "include", "java", "jni"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".c", ".sh", ".java")) or f == "Makefile":
                    src = open(os.path.join(dirpath, f)).read()
                    assert not pat.search(src), os.path.join(dirpath, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert len(re.findall(r"^\s*from oracle import", bench, re.M)) == 1 and "def cpu_baseline" in bench.split("from oracle import")[0][-2500:]


def _check_schedule(u, j, nu, ni, order):
    perm, off = capi.level_schedule(u, j, nu, ni, order)
    n = len(u)
    assert sorted(perm.tolist()) == list(range(n))          # a permutation
    assert off[0] == 0 and off[-1] == n and np.all(np.diff(off) > 0 if n else True)
    level_of = np.empty(n, dtype=np.int64)
    for l in range(len(off) - 1):
        seg = perm[off[l]:off[l + 1]]
        level_of[seg] = l
        # tuples of one level share no user and no item -> they commute
        assert len(set(u[seg].tolist())) == len(seg)
        assert len(set(j[seg].tolist())) == len(seg)
    # per user and per item the levels strictly increase along the CRS order
    for key in (u, j):
        last = {}
        for t in range(n):
            k = int(key[t])
            if k in last:
                assert level_of[t] > last[k]
            last[k] = level_of[t]
    # minimality: level(t) = 1 + max(level of predecessors) (longest dependency chain)
    lu, lj = {}, {}
    for t in range(n):
        want = max(lu.get(int(u[t]), -1), lj.get(int(j[t]), -1)) + 1
        assert level_of[t] == want
        lu[int(u[t])] = lj[int(j[t])] = want
    return perm, off


@pytest.mark.parametrize("order", [0, 1, 2])
def test_level_schedule_small(order):
    d = synth.generate(37, 13, 2, 3, 600, seed=order + 1)
    perm, off = _check_schedule(d.u, d.j, d.n_users, d.n_items, order)
    if order == 0:  # CRS order kept inside a level
        for l in range(len(off) - 1):
            assert np.all(np.diff(perm[off[l]:off[l + 1]]) > 0)


@settings(max_examples=40, deadline=None)
@given(nu=st.integers(1, 9), ni=st.integers(1, 9), n=st.integers(0, 80), seed=st.integers(0, 1000),
       order=st.integers(0, 2))
def test_level_schedule_property(nu, ni, n, seed, order):
    rng = np.random.default_rng(seed)
    u = rng.integers(0, nu, n).astype(np.int32)
    j = rng.integers(0, ni, n).astype(np.int32)
    _check_schedule(u, j, nu, ni, order)


def test_level_schedule_rejects_bad_ids():
    with pytest.raises(capi.CmiError):
        capi.level_schedule(np.array([0, 5], np.int32), np.array([0, 0], np.int32), 3, 2)


def test_level_schedule_zipf_chain():
    """A hot item serialises its tuples: #levels >= its degree."""
    d = synth.generate(200, 50, 1, 2, 3000, seed=4, item_zipf=1.3)
    _, off = capi.level_schedule(d.u, d.j, d.n_users, d.n_items)
    assert len(off) - 1 >= np.bincount(d.j).max()


def test_narrow_runs_properties():
    """Runs partition the levels: every run has >= min_levels levels, all <= max_tuples; levels outside runs are either
    wide or sit in a too-short narrow stretch; launches = runs + lone levels."""
    rng = np.random.default_rng(12)
    for _ in range(50):
        n_levels = int(rng.integers(0, 300))
        sizes = np.where(rng.random(n_levels) < 0.7, rng.integers(1, 257, n_levels), rng.integers(257, 5000, n_levels))
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        run, launches = capi.narrow_runs(off, 256, 16)
        l, expect_launches = 0, 0
        while l < n_levels:
            if run[l] > 0:
                e = l + run[l]
                assert run[l] >= 16 and np.all(sizes[l:e] <= 256) and np.all(run[l + 1:e] == -1)
                assert e == n_levels or sizes[e] > 256                      # maximal to the right
                assert l == 0 or sizes[l - 1] > 256                           # and to the left
                l = e
            else:
                assert run[l] == 0
                l += 1
            expect_launches += 1
        assert launches == expect_launches
        # a narrow stretch that is NOT a run is shorter than min_levels
        l = 0
        while l < n_levels:
            if run[l] == 0 and sizes[l] <= 256:
                e = l
                while e < n_levels and run[e] == 0 and sizes[e] <= 256:
                    e += 1
                assert e - l < 16
                l = e
            else:
                l += 1


def test_conflict_free_blocks_properties():
    rng = np.random.default_rng(13)
    for n_users, n_items, n in ((5, 4, 200), (50, 40, 1000), (1000, 800, 3000), (3, 3, 0)):
        u = rng.integers(0, n_users, n).astype(np.int32)
        j = rng.integers(0, n_items, n).astype(np.int32)
        off = capi.conflict_free_blocks(u, j, n_users, n_items, 64)
        if n == 0:
            assert off.tolist() == [0]
            continue
        assert off[0] == 0 and off[-1] == n and np.all(np.diff(off) > 0) and np.all(np.diff(off) <= 64)
        for b, e in zip(off[:-1], off[1:]):
            assert len(set(u[b:e].tolist())) == e - b and len(set(j[b:e].tolist())) == e - b      # conflict-free
            if e < n and e - b < 64:                                                              # maximal: the next tuple conflicts
                assert u[e] in set(u[b:e].tolist()) or j[e] in set(j[b:e].tolist())

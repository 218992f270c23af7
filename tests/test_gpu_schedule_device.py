"""The hub-chain level schedule built ON THE DEVICE (carskit_amd/csrc/sched_device.hip; what cmi_set_ratings uses from 2 M tuples on) is
the host builder's schedule (level_schedule.cpp build_chain_schedule, property-tested in tests/test_chain_schedule.py) element for
element: same permutation, unit offsets, level offsets and hub side.  The schedule is the order-exact restatement of librec's
MatrixIterator order (CAMF_CI.java:80 `for (MatrixEntry me : trainMatrix)`), so the device build may change where it is computed and
nothing else; so is the spoke arena's position list, checked through a training run that equals the host-scheduled one bit for bit."""
import os

import numpy as np
import pytest

from carskit_amd import capi, synth
from tests import util

pytestmark = pytest.mark.gpu


def _same(u, j, nu, ni, hub, max_chain):
    a = capi.chain_schedule(u, j, nu, ni, hub, max_chain)
    b = capi.chain_schedule_device(u, j, nu, ni, hub, max_chain)
    assert a[3] == b[3]
    for x, y, name in zip(a[:3], b[:3], ("perm", "unit_off", "level_off")):
        assert np.array_equal(x, y), name


@pytest.mark.parametrize("hub", [-1, 0, 1, -2, -3])
@pytest.mark.parametrize("max_chain", [1, 4, 16])
def test_device_schedule_equals_host_small(hub, max_chain):
    rng = np.random.default_rng(100 + hub + max_chain)
    for nu, ni, n in ((1, 1, 1), (7, 5, 30), (40, 300, 2000), (300, 40, 5000), (1000, 1000, 20000)):
        pair = rng.choice(nu * ni, size=min(n, nu * ni), replace=False)
        pair.sort()
        u, j = (pair // ni).astype(np.int32), (pair % ni).astype(np.int32)
        _same(u, j, nu, ni, hub, max_chain)


def test_device_schedule_equals_host_heavy_tail_and_repeats():
    """Zipf items (one long dependency chain through the hottest row) and repeated (user, item) pairs (several contexts of one pair:
    consecutive tuples sharing BOTH rows)."""
    rng = np.random.default_rng(7)
    n, nu, ni = 60000, 5000, 800
    u = np.sort(rng.integers(0, nu, n)).astype(np.int32)
    j = np.minimum(rng.zipf(1.3, n) - 1, ni - 1).astype(np.int32)
    for hub in (-1, 0, 1):
        _same(u, j, nu, ni, hub, 16)
    u2, j2 = np.repeat(u[:20000], 3), np.repeat(j[:20000], 3)
    _same(u2, j2, nu, ni, -1, 16)


def test_device_schedule_equals_host_at_size():
    """5 M tuples (the `small` bench workload's shape): many hub rows per resident lane, the side chosen by unit count."""
    d = synth.generate_fast(100_000, 10_000, 4, 8, 5_000_000)
    for hub in (-3, -2):
        _same(d.u, d.j, d.n_users, d.n_items, hub, 16)


def test_set_ratings_with_device_schedule_trains_bit_identically():
    """cmi_set_ratings with the device-built schedule and arena lists (forced at this size) vs the host-built ones: the same epochs."""
    data = util.small_data(n_users=3000, n_items=400, n_dims=3, conds_per_dim=3, n=60000, seed=71)
    state = synth.init_state("CAMF_CI", data, 64, seed=3, dtype=np.float32)
    runs = []
    for dev in ("0", "1"):
        os.environ["CMI_SCHEDULE_DEVICE"] = dev
        try:
            inst = capi.Instance("CAMF_CI", 64, data.n_users, data.n_items, data.n_conds, flags=capi.FLAG_SCHED_CHAIN | capi.FLAG_SPOKE_ARENA)
            inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, float(data.r.mean()))
            inst.set_ratings(data.u, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
        finally:
            del os.environ["CMI_SCHEDULE_DEVICE"]
        inst.set_states(state)
        losses = [inst.train_epoch(util.LR) for _ in range(3)]
        runs.append((losses, inst.get_states(), inst.schedule_info()))
    assert runs[0][0] == runs[1][0] and runs[0][2] == runs[1][2]
    for name, a in runs[0][1].items():
        assert np.array_equal(a, runs[1][1][name]), name


def test_device_schedule_in_memory_walk_equals_host():
    """Sides with more hub rows than 16 per resident lane (north_star's 10 M users) keep the rows' state in memory (k_walk_mem): forced
    here on a small set."""
    rng = np.random.default_rng(9)
    n, nu, ni = 40000, 6000, 700
    u = np.sort(rng.integers(0, nu, n)).astype(np.int32)
    j = rng.integers(0, ni, n).astype(np.int32)
    os.environ["CMI_SCHED_WALK_MEM"] = "1"
    try:
        for hub in (-1, 0, 1):
            _same(u, j, nu, ni, hub, 16)
    finally:
        del os.environ["CMI_SCHED_WALK_MEM"]

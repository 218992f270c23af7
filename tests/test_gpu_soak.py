"""Concurrency soak (round 6): owner epochs stay bit for bit what they are alone while OTHER work of the same process is in flight on the
GPU -- which is how the reference runs its folds (`cv -p on`: one thread per fold, CARSKit.java:395-412).

Why this file exists: in round 5 an instance whose hottest owners ran as teams was ~2e-7 off the fp64 oracle whenever another owner
epoch ran beside it, and exact alone.  The cause (docs/history/r06.md 1) was a store-data hazard: a record's 16-byte
`buffer_store_dwordx4 ... sN offen` reads its data registers after it has issued, the compiler -- which believes buffer stores with an
SGPR soffset exempt from the gfx9 ">64-bit store data" wait states -- packed the next store's words into the same registers in the next
issue slots, and with a second workgroup's memory traffic on the compute unit the store now and then picked up the NEW low word (right
tags, wrong data: no tag test can see that).  tools/micro/store_data_hazard.hip shows the hazard in isolation; owner_st_words now keeps
the registers untouched for two wait states.  The tests below would have caught it: 23 of 24 repetitions deviated before the fix.

Each configuration trains E epochs from the same injected state REPS times beside a neighbour thread and compares the whole model with
its reference bit for bit: the sequential oracle (order AND arithmetic of CAMF_CI.java:75-131) for the strict fp64 form, the same
instance's lone run for the others (whose arithmetic is the kernels' own; their agreement with the oracle to rounding is
tests/test_gpu_owner.py's subject)."""
import os
import threading

import numpy as np
import pytest

from carskit_amd import capi, synth
from oracle import oracle_c
from tests import util
from tests.test_gpu_owner import _env

pytestmark = pytest.mark.gpu

OWNER, CHAIN, F64, STRICT = capi.FLAG_SCHED_OWNER, capi.FLAG_SCHED_CHAIN, capi.FLAG_STATE_F64, capi.FLAG_STRICT
REPS = int(os.environ.get("CMI_SOAK_REPS", "50"))
EPOCHS = 2


def _instance(model, data, k, flags, state, share, **env):
    def make():
        inst = capi.Instance(model, k, data.n_users, data.n_items, data.n_conds, flags=flags)
        inst.set_hparams(util.REG, util.REG, util.REG, util.REGC, oracle_c.global_mean(data.r))
        if share:
            inst.set_device_share(share)
        inst.set_ratings(data.u, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
        inst.set_states(state)
        return inst
    return _env(make, **env)


class Neighbour:
    """Keeps one kind of other work in flight on the device from a thread of its own until stopped."""

    def __init__(self, kind):
        self.kind, self.stop, self.rounds, self.error = kind, threading.Event(), 0, None
        d = synth.generate(2500, 250, 3, 4, 60000, seed=900, item_zipf=1.2)
        if kind == "owner":      # another instance's persistent owner epoch (both declare the shared device)
            st = synth.init_state("CAMF_CI", d, 64, seed=9, dtype=np.float32)
            self.inst = _instance("CAMF_CI", d, 64, OWNER, st, 2, CMI_OWNER_TEAM="0")
            assert self.inst.schedule_info()["kind"].startswith("owner")
            self.step = lambda: self.inst.train_epoch(util.LR)
        elif kind == "chain":    # another instance's hub-chain level epochs (hundreds of short launches per epoch)
            st = synth.init_state("CAMF_CU", d, 128, seed=9, dtype=np.float32)
            self.inst = _instance("CAMF_CU", d, 128, CHAIN, st, 0)
            assert self.inst.schedule_info()["kind"].startswith("chain")
            self.step = lambda: self.inst.train_epoch(util.LR)
        else:                    # "schedule": the device-side schedule build of another cmi_set_ratings (sched_device.hip's resident walk
            #                      with its parked-wave backstop), checked against the host builder every time
            big = synth.generate_fast(40_000, 4_000, 3, 4, 400_000)
            self.want = capi.chain_schedule(big.u, big.j, big.n_users, big.n_items, -1, 16)

            def build():
                got = capi.chain_schedule_device(big.u, big.j, big.n_users, big.n_items, -1, 16)
                assert got[3] == self.want[3] and all(np.array_equal(x, y) for x, y in zip(got[:3], self.want[:3])), "device schedule differs"
            self.step = build
        self.thread = threading.Thread(target=self.run, daemon=True)

    def run(self):
        try:
            while not self.stop.is_set():
                self.step()
                self.rounds += 1
        except Exception as exc:      # surfaced by the test
            self.error = exc

    def __enter__(self):
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        self.thread.join(timeout=120)
        if getattr(self, "inst", None):
            self.inst.close()


def _final_state(inst, state):
    inst.set_states(state)
    for _ in range(EPOCHS):
        inst.train_epoch(util.LR)
    return inst.get_states()


def _same(a, b):
    return all(np.array_equal(a[n].reshape(b[n].shape), b[n]) for n in b)


CONFIGS = [  # (label, flags, CMI_OWNER_TEAM)
    ("strict fp64, one wavefront per owner", F64 | STRICT, "0"),
    ("fp32, one wavefront per owner", 0, "0"),
    ("fp32, teams", 0, None),
    ("fp64, teams", F64, None),
    ("fp64, every owner a team", F64, "all"),
]


@pytest.mark.parametrize("kind", ["owner", "chain", "schedule"])
@pytest.mark.parametrize("hub", ["item", "user"])
@pytest.mark.parametrize("k", [10, 64, 128])
def test_owner_epochs_beside_other_work_are_bit_identical(kind, hub, k):
    data = synth.generate(1500, 200, 3, 4, 30000, seed=40 + k, item_zipf=1.2)   # (hub = user: the owned rows are the many short ones)
    model = "CAMF_CI" if hub == "item" else "CAMF_CU"
    cases = []
    for label, flags, team in CONFIGS:
        dtype = np.float64 if flags & F64 else np.float32
        state = synth.init_state(model, data, k, seed=5, dtype=dtype)
        inst = _instance(model, data, k, flags | OWNER, state, 2, CMI_OWNER_TEAM=team, CMI_OWNER_HUB=hub, CMI_OWNER_TEAM_MIN=1000)
        info = inst.schedule_info()
        assert info["kind"] == "owner-" + hub
        if team == "all" or (team is None and hub == "item"):   # (automatic teams go to long single-row lists: the hot items)
            assert info["teams"] > 0, (label, info)
        if flags & STRICT:      # the sequential oracle: the reference's order and arithmetic
            st64 = {n: a.astype(np.float64) for n, a in state.items()}
            orc = util.c_oracle(model, data, k, st64, oracle_c.global_mean(data.r), util.REG, util.REG, util.REG, util.REGC)
            for _ in range(EPOCHS):
                orc.epoch(util.LR)
            want = {n: np.asarray(a) for n, a in orc.state.items()}
            assert _same(want, _final_state(inst, state)), label + ": lone run differs from the oracle"
        else:                   # the lone run
            want = _final_state(inst, state)
            assert _same(want, _final_state(inst, state)), label + ": two lone runs differ"
        cases.append((label, inst, state, want))
    with Neighbour(kind) as nb:
        for rep in range(REPS):
            for label, inst, state, want in cases:
                got = _final_state(inst, state)
                assert _same(want, got), "%s, k = %d, hub = %s, beside %s: repetition %d differs (largest deviation %.3e)" % (
                    label, k, hub, kind, rep, max(float(np.max(np.abs(want[n].reshape(got[n].shape).astype(np.float64) - got[n]))) for n in got))
            assert nb.error is None, nb.error
        assert nb.rounds > 0      # the neighbour really was running
    assert nb.error is None, nb.error
    for _, inst, _, _ in cases:
        inst.close()


def test_team_form_beside_owner_epochs_of_other_instances():
    """The round-5 reproducer as a regression test: an fp64 k = 64 instance whose hottest owners are teams, trained beside two other
    owner epochs (three threads, cmi_set_device_share(3)) -- bit for bit the one-wavefront form's model, every time.  (23 of 24 such
    runs deviated by ~2e-7 before the store-data hazard was fixed.)"""
    from concurrent.futures import ThreadPoolExecutor
    ds = [synth.generate(3000, 300, 3, 4, 120000, seed=500 + s, item_zipf=1.2) for s in (1, 2, 3)]

    def make(d, team, flags):
        st = synth.init_state("CAMF_CI", d, 64, seed=5, dtype=np.float64 if flags & F64 else np.float32)
        return _instance("CAMF_CI", d, 64, OWNER | flags, st, 3, CMI_OWNER_TEAM=team)

    for flags in (F64, 0):
        ref = make(ds[0], "0", flags)
        for _ in range(3):
            ref.train_epoch(util.LR)
        want = ref.get_states()
        ref.close()
        for team in (None, "all"):
            for rep in range(6):
                conc = [make(ds[0], team, flags), make(ds[1], "0", flags), make(ds[2], "0", flags)]
                assert conc[0].schedule_info()["teams"] > 0
                with ThreadPoolExecutor(max_workers=3) as pool:
                    list(pool.map(lambda i: [i.train_epoch(util.LR) for _ in range(3)], conc))
                got = conc[0].get_states()
                for c in conc:
                    c.close()
                assert _same(want, got), (flags, team, rep)

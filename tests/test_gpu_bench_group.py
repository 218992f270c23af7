"""`python bench.py --gpus N` WITHOUT torch.distributed.run: one host process, one cmi_group (bench.py bench_group) -- the form the Java /
C++ hosts use (-Dcarskit.shards=N).  On the one-GPU test box both shards sit on device 0 (CMI_BENCH_SHARE_GPU: the in-process exchange),
so this function has run before the first multi-GPU node meets it.  The reference's only parallelism is a thread per
fold (CARSKit.java:395-412); the sharded epoch is this library's own."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("model,shards", [("", 2), ("CAMF_CU", 2), ("", 3)])
def test_bench_group_without_torchrun_on_one_gpu(model, shards):
    env = dict(os.environ, CMI_BENCH_SHARE_GPU="1")
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(v, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(shards), "--steps", "3", "--warmup", "1", "--workload", "small"]
    p = subprocess.run(cmd + (["--model", model] if model else []), capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == shards and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["final_loss"] < rec["first_loss"]
    par = rec["config"]["parallelism"]
    assert par.startswith("one process, cmi_group over %d shards on 1 physical GPU(s)" % shards) and "in-process" in par and "SHARED DEVICE" in par
    sh = rec["config"]["shards"]
    assert len(sh) == shards and all(s["tuples"] == 5_000_000 and s["device"] == 0 and s["exchange"] == "in-process" for s in sh)
    assert [s["user_lo"] for s in sh] == [100_000 * r for r in range(shards)]             # weak scaling: every shard owns its own users
    rf = rec["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1.2 and 0 < rf["frac_whole_step"] <= rf["frac"] * 1.001
    assert len(rf["avg_launch_us"]) == shards and all(x > 0 for x in rf["avg_launch_us"])
    assert rec["compute_ms"] > 0 and rec["exchange_ms"] > 0 and rec["host_ms"] >= 0
    # the three parts cannot exceed the step by more than the concurrency of shards on one device allows
    assert rec["compute_ms"] <= rec["ms_per_step"] * 1.05


@pytest.mark.gpu
def test_bench_group_falls_back_to_the_in_process_exchange_when_rccl_fails():
    """The first multi-GPU node will be the first place RCCL meets more than one rank: if ncclCommInitAll -- or the
    pre-flight exchange behind it -- fails there, the group must still train (in-process peer-copy exchange), say why, and bench.py must
    still print its line.  CMI_GROUP_TRY_RCCL=1 makes the group attempt RCCL although both shards sit on device 0: ncclCommInitAll
    refuses duplicate devices -- a REAL RCCL failure, not a simulated one."""
    env = dict(os.environ, CMI_BENCH_SHARE_GPU="1", CMI_GROUP_TRY_RCCL="1")
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(v, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "small"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    par = rec["config"]["parallelism"]
    assert "FALLBACK: ncclCommInitAll failed" in par and rec["config"]["exchange"] == "in-process", par
    assert "[cmi] group: in-process exchange (peer copies) -- FALLBACK" in p.stderr
    assert rec["value"] > 0 and rec["final_loss"] < rec["first_loss"]


@pytest.mark.gpu
def test_group_fallback_trains_exactly_like_the_plain_in_process_group():
    """Same data, same state: the group that fell back from RCCL and the group that never tried it run the same exchange, bit for bit."""
    from carskit_amd import capi, synth
    from tests import util
    data = synth.generate(2000, 300, 3, 4, 60000, seed=31)
    state = synth.init_state("CAMF_CI", data, 32, seed=4, dtype=np.float32)
    outs = []
    for tryit in (None, "1"):
        if tryit: os.environ["CMI_GROUP_TRY_RCCL"] = tryit
        try:
            g = capi.Group("CAMF_CI", 32, data.n_users, data.n_items, data.n_conds, 2, devices=[0, 0])
            g.set_hparams(util.REG, util.REG, util.REG, util.REGC, 3.0)
            g.set_ratings(data.u, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
        finally:
            os.environ.pop("CMI_GROUP_TRY_RCCL", None)
        assert ("FALLBACK" in g.exchange_path()) == bool(tryit), g.exchange_path()
        g.set_states(state)
        losses = [g.train_epoch(util.LR) for _ in range(3)]
        outs.append((losses, g.get_states()))
        g.close()
    assert outs[0][0] == outs[1][0]
    for n in outs[0][1]:
        assert np.array_equal(outs[0][1][n], outs[1][1][n]), n


def test_merge_user_parts_builds_one_context_table():
    """Every part numbers its context combinations in its own first-seen order; the merged set has ONE table and every tuple keeps
    its condition list (bench_group fed part 0's table to every part before: VERDICT r4)."""
    from carskit_amd import synth
    parts = [synth.generate_fast(300, 50, 3, 3, 2000, seed=5 + 1000 * r) for r in range(3)]
    assert any(not np.array_equal(parts[0].ctx_conds, p.ctx_conds) for p in parts[1:])     # the parts' tables do differ
    m = synth.merge_user_parts(parts)
    assert m.n_users == sum(p.n_users for p in parts) and m.n == sum(p.n for p in parts) and m.n_conds == parts[0].n_conds
    rows = {m.ctx_conds[m.ctx_ptr[c]:m.ctx_ptr[c + 1]].tobytes() for c in range(m.n_ctx)}
    assert len(rows) == m.n_ctx                                                            # no combination twice
    off = base = 0
    for p in parts:
        conds_p = p.ctx_conds.reshape(p.n_ctx, -1)[p.ctx]
        conds_m = m.ctx_conds.reshape(m.n_ctx, -1)[m.ctx[off:off + p.n]]
        assert np.array_equal(conds_p, conds_m)
        assert np.array_equal(m.u[off:off + p.n], p.u + base) and np.array_equal(m.j[off:off + p.n], p.j)
        off, base = off + p.n, base + p.n_users
    first = np.unique(m.ctx, return_index=True)[1]
    assert np.all(np.diff(first) > 0)                                                       # ids in first-seen order of the merged stream

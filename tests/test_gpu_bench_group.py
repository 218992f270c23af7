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

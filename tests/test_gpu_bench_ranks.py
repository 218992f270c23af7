"""The N>1 branch of bench.py (one process per rank, ShardedEpochRunner, global mean / loss / timing all-reduces) run as the
driver runs it -- `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` -- but with both ranks on
the single GPU of the test box over gloo (CMI_BENCH_SHARE_GPU), so the code path is exercised before the 8-GPU run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("model", ["", "CAMF_CU", "CAMF_CUCI"])   # models without userBias crashed the N>1 branch in round 1
def test_bench_two_ranks_share_one_gpu(model):
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, CMI_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "small"] + (["--model", model] if model else [])
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                      # rank 0 prints exactly one JSON line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["roofline"]["bound"] == "hbm" and "cpu_baseline" not in rec     # the CPU leg is rank 0 at N=1 only
    assert rec["config"]["parallelism"].startswith("user-sharded x2") and "mean merge" in rec["config"]["parallelism"]
    assert rec["config"]["final_loss"] < rec["config"]["first_loss"]            # the merged run converges
    assert "5000000 ratings per GPU" in rec["config"]["workload"]                # weak scaling: per-GPU work is fixed
    # N > 1: the job's own roofline (all ranks' schedule bytes over the slowest rank's epoch vs N x 8 TB/s) and the step's split
    assert rec["roofline_aggregate"]["peak"] == 16000.0 and 0 < rec["roofline_aggregate"]["frac_whole_step"] <= rec["roofline_aggregate"]["frac"]
    assert rec["compute_ms"] > 0 and len(rec["rank_compute_ms"]) == 2 and len(rec["rank_avg_launch_us"]) == 2
    assert rec["exchange_ms"] is None          # gloo: torch issues the collectives (the library times its own RCCL exchange only)

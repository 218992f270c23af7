"""usage (GPU box): python tools/exp/fm_comm_sweep_time.py [sweeps]  -- the FM sweep of the C4 share with its per-phase exchange issued
by the library over RCCL at WORLD SIZE 1 (cmi_fm_comm_sweep: reduce -> ncclAllReduce of [num | den] -> apply for the w0 / item / context
phases, user phases fused) against the fused single-GPU sweep (cmi_fm_sweep): what the ~130 collectives of a sweep cost in launches and
stream ordering when the wire itself costs nothing (one rank).  The N > 1 cost adds the xGMI latency per collective on top."""
import os, socket, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch
import torch.distributed as tdist
from carskit_amd import capi, synth, dist as cdist

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
k = 64
data = synth.generate_fast(625_000, 500_000, 4, 16, 25_000_000)
p = data.n_users + data.n_items + data.n_conds
rng = np.random.default_rng(1)
w, V = rng.random(p), 0.1 * rng.standard_normal((p, k))


def make():
    g = capi.FMInstance(k, data.n_users, data.n_items, data.n_conds, data.n_dims)
    g.set_hparams(synth.java_float(0.01), synth.java_float(0.02))
    g.set_ratings(data.u, data.j, data.ctx, data.r)
    g.set_model(0.0, w, V)
    g.init()
    return g


def timed(fn, g):
    fn()
    g.synchronize()
    t0 = time.perf_counter()
    for _ in range(sweeps):
        fn()
    g.synchronize()
    return (time.perf_counter() - t0) / sweeps * 1e3


a = make()
fused = timed(a.sweep, a)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
b = make()
run = cdist.ShardedFMRunner(cdist.GpuFMEngine(b, 0), tdist, always_exchange=True)
assert run.lib_comm
comm = timed(run.sweep, b)
n_coll = 1 + 2 + 2 * k
print("FM C4 share, k=%d: fused sweep %.2f ms; sweep with %d per-phase RCCL all-reduces at world size 1 %.2f ms: +%.2f ms = %.1f us per exchanged phase"
      % (k, fused, n_coll, comm, comm - fused, (comm - fused) * 1e3 / n_coll))
tdist.destroy_process_group()

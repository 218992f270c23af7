"""Multi-GPU host layer: user-sharded data-parallel epochs with an item-side delta exchange.

The reference path is one sequential loop (no collective exists to translate).  The algorithm shards
naturally BY USER: P[u], userBias[u], ucBias[u,*] are then touched by exactly one rank, while the
item-side containers (Q, itemBias, icBias) are replicated.  One epoch =

    every rank:  local SGD epoch over its own tuples (order-exact level schedule, cmi_train_epoch_async)
    exchange:    delta_r = itemside_r - itemside_at_epoch_start ;  reduce-scatter + all-gather (sum) of one flat bucket ;
                 itemside = itemside_at_epoch_start + (sum_r delta_r) / W    (RCCL over xGMI; "mean" merge, see ShardedEpochRunner)
    loss:        all-reduce(sum) of the per-rank epoch losses (feeds the host's bold-driver / isConverged)

With world_size == 1 this is exactly the single-GPU path.  With more ranks it is NOT the reference's
sequential semantics (W local SGD streams merged per epoch), so multi-GPU accuracy is reported as an RMSE
band against the 1-GPU result; the 1-GPU order-exact path stays the parity anchor.

The runner is written against a tiny engine protocol so the exchange algebra can be tested on CPU with
torch.distributed's gloo backend (tests/test_dist_gloo.py drives it with the CPU oracle as the engine);
the product engine is GpuEngine over a capi.Instance and there is no CPU engine in this package.
"""
import numpy as np

from . import capi

ITEM_SIDE = {
    "BiasedMF": ("Q", "itemBias"),
    "PMF": ("Q",),
    "CAMF_CI": ("Q", "icBias"),
    "CAMF_CU": ("Q", "itemBias"),
    "CAMF_CUCI": ("Q", "icBias"),
    # CAMF_C also shares condBias; it is serial-only on one GPU and is not sharded
}


class _DevArray:
    """Expose a raw device pointer through __cuda_array_interface__ so torch can alias it (zero copy)."""

    def __init__(self, ptr, count, np_dtype):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": np.dtype(np_dtype).str,
                                         "data": (int(ptr), False), "version": 2, "strides": None}


class GpuEngine:
    """Engine over one capi.Instance.  The exchange runs entirely on the instance's HIP stream: two hand-written elementwise
    kernels (cmi_exchange_pack / cmi_exchange_apply) around the collective -- local epoch -> pack -> reduce-scatter + all-gather ->
    apply -> loss all-reduce are ordered on the device, and the host synchronises once per epoch, when it reads the loss it needs
    for isConverged().

    Who issues the collective: with the `nccl` backend (RCCL; a real multi-GPU job) the LIBRARY does, through cmi_comm_* -- the one
    exchange implementation the single-process hosts use too (cmi_group_*, group_api.cpp exchange_collective); torch.distributed
    only hands the 128-byte RCCL unique id from rank 0 to the other ranks (`lib_comm`).  With `gloo` (the CPU / shared-GPU tests) torch
    issues it on the bucket tensor, with the instance's stream made current."""

    def __init__(self, inst, device_index, world=1, dist=None, group=None):
        import torch
        self.inst = inst
        self.torch = torch
        self.device = torch.device("cuda", device_index)
        self.lib_comm = False
        self.exchange_path = "torch.distributed collectives on the instance's stream"
        import os
        if dist is not None and dist.is_initialized() and dist.get_backend(group) == "nccl" and not os.environ.get("CMI_DIST_TORCH"):
            # (CMI_DIST_TORCH=1: let torch issue the collectives even under RCCL -- the A/B of the two issuers of the same exchange)
            rank = dist.get_rank(group)
            idt = torch.zeros(capi.COMM_ID_BYTES, dtype=torch.uint8, device=self.device)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
            if dist.get_world_size(group) > 1:
                dist.broadcast(idt, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            # ncclCommInitRank is itself a collective: a rank that failed BEFORE it would leave the others blocked inside it.  So the
            # part that can fail locally (cmi_exchange_setup: model not shardable, no memory for bucket + snapshot) runs first and the
            # ranks vote on it (a cheap all-reduce); only if every rank is ready do ALL of them create the library's communicator,
            # otherwise ALL use the torch-issued form of the same exchange -- said loudly on stderr and recorded in `exchange_path`.
            ok = torch.ones(1, device=self.device)
            try:
                inst.exchange_setup(pad_to=max(1, world))
            except Exception as e:   # noqa: BLE001
                import sys
                print("carskit_amd.dist: cmi_exchange_setup failed on rank %d (%s); all ranks use torch.distributed collectives" % (rank, e),
                      file=sys.stderr, flush=True)
                ok.zero_()
            if dist.get_world_size(group) > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if float(ok.item()) > 0:
                # (a failure inside ncclCommInitRank itself is fatal for the job: RCCL's own timeout / abort ends it, there is no
                # fallback a single rank could take without the others)
                inst.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, dist.get_world_size(group))   # also snapshots the item-side state
                self.lib_comm = True
                self.exchange_path = "library (cmi_comm_*: RCCL reduce-scatter + all-gather issued by libcarskit_mi355x)"
                return
        ptr, cnt, dt = inst.exchange_setup(pad_to=max(1, world))   # also snapshots the item-side state
        self.bucket = torch.as_tensor(_DevArray(ptr, cnt, dt), device=self.device)
        self.loss = torch.as_tensor(_DevArray(inst.loss_device_ptr(), 1, np.float64), device=self.device)
        self.ext = torch.cuda.ExternalStream(inst.stream_ptr(), device=self.device)

    def start_epoch(self, lr):
        self.inst.train_epoch_async(lr)

    def local_loss(self):
        return self.inst.last_loss()          # synchronises the instance stream

    def comm_epoch_tail(self, scale):
        """lib_comm: the exchange behind the epoch start_epoch() enqueued, and the global loss (the epoch's one host sync)."""
        self.inst.comm_exchange(scale)
        return self.inst.last_loss()

    def exchange_stream(self):
        return self.torch.cuda.stream(self.ext)

    def pack(self):
        self.inst.exchange_pack()
        return self.bucket

    def apply(self, scale):
        self.inst.exchange_apply(scale)

    def loss_tensor(self):
        return self.loss

    def finish(self):
        self.inst.synchronize()


class TorchEngineMixin:
    """pack / apply for engines whose item-side state is a dict of torch tensors (the CPU test engines): the same algebra as the
    HIP kernels, in torch ops."""

    def _setup_exchange(self, world):
        import torch
        item = self.item_state()
        self._names = list(item)
        total = sum(t.numel() for t in item.values())
        total = (total + world - 1) // world * world
        any_t = next(iter(item.values()))
        self._bucket = torch.zeros(total, dtype=any_t.dtype, device=any_t.device)
        self._snap = {n: t.clone() for n, t in item.items()}
        self._loss = torch.zeros(1, dtype=torch.float64, device=any_t.device)

    def start_epoch(self, lr):
        self._loss[0] = self.epoch_local(lr)

    def local_loss(self):
        return float(self._loss.item())

    def exchange_stream(self):
        import contextlib
        return contextlib.nullcontext()

    def pack(self):
        import torch
        off = 0
        for n in self._names:
            x = self.item_state()[n].view(-1)
            torch.sub(x, self._snap[n].view(-1), out=self._bucket[off:off + x.numel()])
            off += x.numel()
        return self._bucket

    def apply(self, scale):
        off = 0
        for n in self._names:
            x = self.item_state()[n].view(-1)
            x.copy_(self._snap[n].view(-1) + scale * self._bucket[off:off + x.numel()])
            self._snap[n].view(-1).copy_(x)
            off += x.numel()

    def loss_tensor(self):
        return self._loss

    def finish(self):
        pass


MERGE_RULES = ("mean", "sum")


class ShardedEpochRunner:
    """One global epoch = every rank's local order-exact epoch over its users' ratings, then
        item_side = start + scale * sum_r (item_side_r - start),   scale = 1/W ("mean", default) or 1 ("sum").
    Which rule: tests/exp_merge_rule.py (table in DESIGN.md section 7) -- with W stale local passes the SUM of the moves
    overshoots as soon as a rank's pass moves an item row a good part of the way to its optimum (the bench's weak-scaling shape:
    500 ratings per item per rank at lr 0.02), the loss starts to oscillate and the bold driver collapses the rate; the MEAN never
    increased the loss in any configuration tried and stays closest to the sequential (W = 1) result for W >= 4."""

    def __init__(self, engine_or_inst, dist, device_index=None, group=None, always_exchange=False, merge="mean"):
        import torch
        if merge not in MERGE_RULES:
            raise ValueError("merge must be one of %s" % (MERGE_RULES,))
        self.torch = torch
        self.dist = dist
        self.group = group
        self.merge = merge
        self.world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
        if isinstance(engine_or_inst, capi.Instance):
            if device_index is None:
                device_index = torch.cuda.current_device()
            engine_or_inst = GpuEngine(engine_or_inst, device_index, self.world, dist, group)
        self.engine = engine_or_inst
        if hasattr(self.engine, "_setup_exchange"):
            self.engine._setup_exchange(self.world)
        self.always_exchange = always_exchange and dist is not None and dist.is_initialized()
        import os
        backend = dist.get_backend(group) if dist is not None and dist.is_initialized() else ""
        # reduce-scatter + all-gather of the bucket (each rank's slice of a 1 GB Q goes straight to its owner and back: 2 x S/W per
        # link pair on the fully connected xGMI mesh) where the backend has them (RCCL); gloo (CPU tests) only has all-reduce
        # (in-place forms: this rank's slice of the bucket is both the reduce-scatter output and the all-gather input)
        self.rs_ag = backend == "nccl" and (self.world > 1 or self.always_exchange) and not os.environ.get("CMI_DIST_ALLREDUCE")

    def epoch(self, lr):
        """One global epoch; returns the global loss (sum over ranks)."""
        dist, eng = self.dist, self.engine
        eng.start_epoch(lr)
        if self.world == 1 and not self.always_exchange:
            return eng.local_loss()
        scale = 1.0 / self.world if self.merge == "mean" else 1.0
        if getattr(eng, "lib_comm", False):      # RCCL: the library issues the collectives itself (cmi_comm_*)
            return eng.comm_epoch_tail(scale)
        with eng.exchange_stream():
            bucket = eng.pack()
            if self.rs_ag:
                shard = bucket.view(self.world, -1)[dist.get_rank(self.group)]
                dist.reduce_scatter_tensor(shard, bucket, op=dist.ReduceOp.SUM, group=self.group)
                dist.all_gather_into_tensor(bucket, shard, group=self.group)
            else:
                dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group)
            eng.apply(scale)
            lt = eng.loss_tensor()
            dist.all_reduce(lt, op=dist.ReduceOp.SUM, group=self.group)
            total = float(lt.item())    # read back on the stream the collectives were ordered on: the one host sync of the epoch
        eng.finish()
        return total


def preflight_exchange(make_runner, item_state, dist, group=None, lr=0.02, epochs=1):
    """Before a timed multi-rank run (bench.py --gpus N under torch.distributed.run): the first collective the job ever issues must not be
    a timed one, and a broken exchange must be noticed before it is measured.  `make_runner(force_torch)` builds a ShardedEpochRunner over
    a SMALL problem from a fixed state -- force_torch False: the PRIMARY exchange, the one the job would use (under the `nccl` backend the
    library's own RCCL communicator, cmi_comm_*); True: the same exchange issued through torch.distributed on the bucket tensor -- and
    `item_state(runner)` returns {name: numpy array} of the replicated item-side containers.  One epoch + exchange through each, then:
      * every rank holds the SAME item-side state after the primary exchange (digests all-gathered): what a sum / all-gather guarantees;
      * primary == torch-issued: bit for bit at world 2 (a + b commutes, the mean scales both alike), to 1e-5 relative beyond (the
        association of W > 2 floating-point additions is the collective's own).
    Verdicts are the same on every rank (votes: one all-reduce each).  Returns {"ok", "verified", "note", "loss"}:
      ok False        the primary exchange raised somewhere, or left the ranks with different states: the caller runs the torch-issued
                      form instead (CMI_DIST_TORCH=1) and says so;
      ok True, verified False   the primary exchange is consistent across ranks but the torch-issued comparison could not be made or
                      disagreed with it: the caller keeps the primary exchange and prints the note;
      ok True, verified True    both agree.
    A hang inside a collective cannot be caught here: RCCL's own watchdog / timeout ends such a job."""
    import hashlib
    import torch
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"

    def vote(good):
        t = torch.tensor([1.0 if good else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return float(t.item()) == 1.0

    note, primary, loss = "", None, float("nan")
    try:
        run = make_runner(False)
        for _ in range(epochs):
            loss = run.epoch(lr)
        primary = {n: np.array(a, copy=True) for n, a in item_state(run).items()}
    except Exception as e:   # noqa: BLE001 -- whatever the exchange raised on this rank
        note = "primary exchange raised %r" % (e,)
    if not vote(not note):
        return {"ok": False, "verified": False, "note": note or "the primary exchange failed on another rank", "loss": loss}
    digest = int.from_bytes(hashlib.sha256(b"".join(np.ascontiguousarray(primary[n]).tobytes() for n in sorted(primary))).digest()[:7], "little")
    mine = torch.tensor([digest], dtype=torch.int64, device=dev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    if len({int(t.item()) for t in every}) != 1:     # (the same list on every rank: no vote needed)
        return {"ok": False, "verified": False, "note": "ranks hold different item-side states after the primary exchange", "loss": loss}
    try:
        run_t = make_runner(True)
        for _ in range(epochs):
            loss_t = run_t.epoch(lr)
        other = item_state(run_t)
        for n in sorted(primary):
            a, b = primary[n], np.asarray(other[n])
            same = np.array_equal(a, b) if world <= 2 else np.allclose(a, b, rtol=1e-5, atol=1e-7)
            if not same:
                note = "MISMATCH: container %s differs from the torch-issued exchange (largest deviation %.3e)" % (n, float(np.max(np.abs(a - b))))
                break
        if not note and not (loss == loss_t if world <= 2 else abs(loss - loss_t) <= 1e-9 * abs(loss_t)):
            note = "MISMATCH: global loss %r differs from the torch-issued exchange's %r" % (loss, loss_t)
    except Exception as e:   # noqa: BLE001
        note = "torch-issued comparison run raised %r" % (e,)
    verified = vote(not note)
    return {"ok": True, "verified": verified, "note": note or ("" if verified else "the comparison failed on another rank"), "loss": loss}


def shard_by_user(data, rank, world):
    """Tuples of the users owned by `rank`: contiguous user-id ranges cut so that every rank gets about the same
    number of ratings; CRS order is kept; user ids are re-based to the shard (local id = global id - start).
    Returns (RatingData shard, (user_start, user_end))."""
    from .synth import RatingData
    deg = np.bincount(data.u, minlength=data.n_users).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(deg)])
    if data.n_users < world:
        raise ValueError("fewer users than ranks")
    cuts = [0]
    for r in range(1, world):
        c = int(np.searchsorted(cum, data.n * r / world, side="left"))
        cuts.append(min(max(c, cuts[-1] + 1), data.n_users - (world - r)))  # >= 1 user per rank
    cuts.append(data.n_users)
    lo, hi = cuts[rank], cuts[rank + 1]
    idx = np.flatnonzero((data.u >= lo) & (data.u < hi))
    return (RatingData(hi - lo, data.n_items, data.n_conds, data.n_dims, (data.u[idx] - lo).astype(np.int32),
                       data.j[idx], data.ctx[idx], data.r[idx], data.ctx_ptr, data.ctx_conds, data.min_rate,
                       data.max_rate, dict(data.meta)), (lo, hi))


def global_mean(dist, r, device=None):
    """SparseMatrix.getGlobalAvg over the union of all ranks' training tuples (sum / #non-zero)."""
    import torch
    v = torch.tensor([float(np.sum(r)), float(np.count_nonzero(r))], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(v)
    return float(v[0].item() / v[1].item())


# ---- FM (ALS sweep): ratings sharded by user, per-phase exchange of (num, den) --------------------------------

class GpuFMEngine:
    """Engine over one capi.FMInstance holding this rank's ratings (users re-based to the shard)."""

    def __init__(self, fm_inst, device_index):
        import os
        import torch
        self.inst, self.torch = fm_inst, torch
        self.device = torch.device("cuda", device_index)
        self._views = {}
        # The ~130 exchanged phases of a sweep are each a few hundred microseconds long: a host round trip before and after
        # every collective would cost as much as the phases.  The instance's own HIP stream is therefore made torch's current
        # stream for the collective, which orders  reduce kernel -> all-reduce -> apply kernel  on the device with no host
        # synchronisation at all.  CMI_DIST_SYNC=1 restores the blocking form.
        self.ext = None if os.environ.get("CMI_DIST_SYNC") else torch.cuda.ExternalStream(fm_inst.stream_ptr(), device=self.device)

    def num_phases(self):
        return self.inst.num_phases()

    def phase_reduce(self, ph):
        self.inst.phase_reduce(ph)

    def phase_apply(self, ph):
        self.inst.phase_apply(ph)

    def phase_run(self, ph):
        self.inst.phase_run(ph)

    def phase_tensor(self, ph):
        ptr, cnt = self.inst.phase_buffer(ph)
        key = (ptr, cnt)
        if key not in self._views:
            self._views[key] = self.torch.as_tensor(_DevArray(ptr, cnt, np.float64), device=self.device)
        return self._views[key]

    def before_exchange(self):
        if self.ext is None:
            self.inst.synchronize()

    def after_exchange(self):
        if self.ext is None:
            self.torch.cuda.synchronize(self.device)

    def exchange_stream(self):
        """Context manager under which the runner issues the collective."""
        import contextlib
        return self.torch.cuda.stream(self.ext) if self.ext is not None else contextlib.nullcontext()


def fm_phase_field(ph):
    """-1: w0; 0 users, 1 items, 2 context features (phase numbering of cmi_fm_num_phases)."""
    if ph == 0:
        return -1
    return (ph - 1) if ph < 4 else (ph - 4) % 3


class ShardedFMRunner:
    """One FM sweep over user-sharded ratings (reference FM.java:148-218 per phase): every rank reduces its local
    numerator/denominator sums; phases whose coordinates are shared by all ranks (w0, items, context features) sum
    them with an all-reduce before the update, user phases stay local (a user's ratings live on one rank).
    Every rank then applies the same update, so the replicated item/context part of the model stays identical."""

    def __init__(self, engine, dist, group=None, always_exchange=False):
        self.engine, self.dist, self.group = engine, dist, group
        self.world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
        # tests: run the exchange path (collective included) even at world size 1
        self.always_exchange = always_exchange and dist is not None and dist.is_initialized()
        # RCCL (a real multi-GPU job): the library issues the ~130 per-phase all-reduces of a sweep itself (cmi_fm_comm_sweep) -- no
        # Python between the phases; torch.distributed only hands the RCCL unique id to the ranks.  gloo (tests): torch issues them.
        self.lib_comm = False
        if (isinstance(engine, GpuFMEngine) and dist is not None and dist.is_initialized() and dist.get_backend(group) == "nccl"
                and (self.world > 1 or self.always_exchange)):
            import torch
            rank = dist.get_rank(group)
            idt = torch.zeros(capi.COMM_ID_BYTES, dtype=torch.uint8, device=engine.device)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
            if self.world > 1:
                dist.broadcast(idt, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            engine.inst.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, self.world)
            self.lib_comm = True

    def sweep(self):
        eng, dist = self.engine, self.dist
        if self.lib_comm:
            eng.inst.comm_sweep()
            return
        for ph in range(eng.num_phases()):
            exchange = (self.world > 1 or self.always_exchange) and fm_phase_field(ph) != 0
            if not exchange and hasattr(eng, "phase_run"):
                eng.phase_run(ph)       # nothing to exchange: reduce + update in one pass
                continue
            eng.phase_reduce(ph)
            if exchange:
                if hasattr(eng, "before_exchange"):
                    eng.before_exchange()
                if hasattr(eng, "exchange_stream"):
                    with eng.exchange_stream():
                        dist.all_reduce(eng.phase_tensor(ph), op=dist.ReduceOp.SUM, group=self.group)
                else:
                    dist.all_reduce(eng.phase_tensor(ph), op=dist.ReduceOp.SUM, group=self.group)
                if hasattr(eng, "after_exchange"):
                    eng.after_exchange()
            eng.phase_apply(ph)

"""Multi-GPU host layer: user-sharded data-parallel epochs with an item-side delta exchange.

The reference path is one sequential loop (no collective exists to translate).  The algorithm shards
naturally BY USER: P[u], userBias[u], ucBias[u,*] are then touched by exactly one rank, while the
item-side containers (Q, itemBias, icBias) are replicated.  One epoch =

    every rank:  local SGD epoch over its own tuples (order-exact level schedule, cmi_train_epoch_async)
    exchange:    delta_r = itemside_r - itemside_at_epoch_start ;  all-reduce(sum) ;
                 itemside = itemside_at_epoch_start + sum_r delta_r          (one flat bucket, RCCL over xGMI)
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
    """Engine over one capi.Instance: the item-side containers are aliased as torch tensors in place."""

    def __init__(self, inst, device_index):
        import torch
        self.inst = inst
        self.torch = torch
        self.device = torch.device("cuda", device_index)
        self.item = {}
        for name in ITEM_SIDE[inst.model]:
            ptr, cnt, dt = inst.state_device_ptr(name)
            self.item[name] = torch.as_tensor(_DevArray(ptr, cnt, dt), device=self.device)
        # One exchange per epoch: two host synchronisations around it cost nothing next to a 25 ms epoch, so the blocking
        # form is the default here.  CMI_DIST_STREAM=1 issues the exchange with the instance's HIP stream as torch's current
        # stream instead (device-side ordering, as the FM runner does for its ~130 exchanges per sweep).
        import os
        self.ext = torch.cuda.ExternalStream(inst.stream_ptr(), device=self.device) if os.environ.get("CMI_DIST_STREAM") else None

    def epoch_local(self, lr):
        self.inst.train_epoch_async(lr)
        loss = self.inst.last_loss()          # synchronises the instance stream
        return loss

    def item_state(self):
        return self.item

    def before_exchange(self):
        if self.ext is None:
            self.inst.synchronize()               # kernels of the epoch are done before torch touches the state

    def after_exchange(self):
        if self.ext is None:
            self.torch.cuda.synchronize(self.device)  # exchange finished before the next epoch's kernels start

    def exchange_stream(self):
        import contextlib
        return self.torch.cuda.stream(self.ext) if self.ext is not None else contextlib.nullcontext()


class ShardedEpochRunner:
    def __init__(self, engine_or_inst, dist, device_index=None, group=None, always_exchange=False):
        import torch
        self.torch = torch
        self.dist = dist
        self.group = group
        if isinstance(engine_or_inst, capi.Instance):
            if device_index is None:
                device_index = torch.cuda.current_device()
            engine_or_inst = GpuEngine(engine_or_inst, device_index)
        self.engine = engine_or_inst
        item = self.engine.item_state()
        self.names = list(item)
        total = sum(t.numel() for t in item.values())
        any_t = next(iter(item.values()))
        self.bucket = torch.empty(total, dtype=any_t.dtype, device=any_t.device)
        self.start = {n: t.clone() for n, t in item.items()}   # item-side state at epoch start
        self.world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
        self.always_exchange = always_exchange and dist is not None and dist.is_initialized()

    def epoch(self, lr):
        """One global epoch; returns the global loss (sum over ranks)."""
        torch, dist = self.torch, self.dist
        loss = self.engine.epoch_local(lr)
        if self.world == 1 and not self.always_exchange:
            return loss
        if hasattr(self.engine, "before_exchange"):
            self.engine.before_exchange()
        import contextlib
        ctx = self.engine.exchange_stream() if hasattr(self.engine, "exchange_stream") else contextlib.nullcontext()
        with ctx:
            item = self.engine.item_state()
            off = 0
            for n in self.names:
                x = item[n].view(-1)
                torch.sub(x, self.start[n].view(-1), out=self.bucket[off:off + x.numel()])
                off += x.numel()
            dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group)
            off = 0
            for n in self.names:
                x = item[n].view(-1)
                torch.add(self.start[n].view(-1), self.bucket[off:off + x.numel()], out=x)
                self.start[n].view(-1).copy_(x)
                off += x.numel()
            lt = torch.tensor([loss], dtype=torch.float64, device=self.bucket.device)
            dist.all_reduce(lt, op=dist.ReduceOp.SUM, group=self.group)
            total = float(lt.item())    # read back on the same stream the all-reduce was ordered on
        if hasattr(self.engine, "after_exchange"):
            self.engine.after_exchange()
        return total


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

    def sweep(self):
        eng, dist = self.engine, self.dist
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

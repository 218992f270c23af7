"""FM (reference FM.java ALS sweep) on the GPU vs the dense CPU oracle.  fp64 on both sides; the GPU never
stores errors[] / Q[][] (an error is err0 + the running delta sums of its coordinates), adds a coordinate's sums per
piece of its support and uses size*reg for the denominator's regulariser instead of the Java's one-add-per-rating,
so equality is to rounding: model 1e-8 relative, predictions/RMSE 1e-9."""
import numpy as np
import pytest

from carskit_amd import capi, synth
from oracle import oracle_c
from tests import util
from tests.test_oracle_fm import REGLF, REGLW, fm_init_model

pytestmark = pytest.mark.gpu


DET = 0                             # the DEFAULT since round 6: sums in an order the layout alone decides (bit-reproducible, FM.java:148-218)
RELAXED = capi.FM_FLAG_RELAXED_SUMS # the opt-in: a record's products are added with LDS atomics as it is evaluated (order varies run to run)
FORM = 0                            # the form make_fm() builds unless a test passes flags


def make_fm(data, k, seed, flags=None):
    w0, w, V = fm_init_model(data.n_users, data.n_items, data.n_conds, k, seed)
    orc = oracle_c.FMOracle(k, data.n_users, data.n_items, data.n_conds, data.n_dims, data.u, data.j, data.ctx,
                            data.r, w0, w, V, REGLW, REGLF)
    g = capi.FMInstance(k, data.n_users, data.n_items, data.n_conds, data.n_dims, flags=FORM if flags is None else flags)
    g.set_hparams(REGLW, REGLF)
    g.set_ratings(data.u, data.j, data.ctx, data.r)
    g.set_model(w0, w, V)
    return orc, g


@pytest.mark.parametrize("flags", [DET, RELAXED])
@pytest.mark.parametrize("k", [1, 4, 64, 70])
def test_fm_sweeps_match_oracle(k, flags):
    data = util.small_data(n_users=40, n_items=15, n_dims=2, conds_per_dim=3, n=500, seed=51)
    assert data.n_ctx > data.n_conds or data.n_ctx > 0
    orc, g = make_fm(data, k, 3, flags)
    orc.init()
    g.init()
    for it in range(3):
        orc.sweep()
        g.sweep()
        w0, w, V = g.get_model()
        assert abs(w0 - orc.w0) <= 1e-10 * max(1.0, abs(orc.w0)), it
        np.testing.assert_allclose(w, orc.w, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(V, orc.V, rtol=1e-8, atol=1e-11)
    want = np.array([orc.predict(int(u), int(j), int(c)) for u, j, c in zip(data.u, data.j, data.ctx)])
    got = g.predict(data.u, data.j, data.ctx)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
    rmse_o = np.sqrt(np.mean((data.r - np.clip(want, 1, 5)) ** 2))
    rmse_g = np.sqrt(np.mean((data.r - g.predict(data.u, data.j, data.ctx, bound=(1.0, 5.0))) ** 2))
    assert abs(rmse_o - rmse_g) <= 1e-9


def test_fm_train_equals_init_plus_sweeps_and_split_phases():
    data = util.small_data(n_users=60, n_items=20, n_dims=3, conds_per_dim=2, n=900, seed=52)
    _, a = make_fm(data, 8, 4, DET)
    _, b = make_fm(data, 8, 4, DET)
    a.train(2)
    b.init()
    for _ in range(2):      # reduce/apply split (what a multi-GPU host drives) == fused sweep
        for ph in range(b.num_phases()):
            b.phase_reduce(ph)
            b.phase_apply(ph)
    b.synchronize()
    wa, wb = a.get_model(), b.get_model()
    assert wa[0] == wb[0] and np.array_equal(wa[1], wb[1]) and np.array_equal(wa[2], wb[2])
    assert b.num_phases() == 4 + 3 * 8
    ptr, cnt = b.phase_buffer(2)
    assert ptr and cnt == 2 * data.n_items


def test_fm_edge_cases():
    data = util.small_data(n_users=10, n_items=5, n=60, seed=53)
    g = capi.FMInstance(4, data.n_users, data.n_items, data.n_conds, data.n_dims)
    with pytest.raises(capi.CmiError):
        g.init()                                            # nothing set yet
    with pytest.raises(capi.CmiError):
        g.set_ratings(data.u + 99, data.j, data.ctx, data.r)
    # a rating whose context id >= numConditions simply has no context feature
    orc, g2 = make_fm(data.subset(np.arange(5)), 4, 1)
    orc.init()
    g2.init()
    orc.sweep()
    g2.sweep()
    np.testing.assert_allclose(g2.get_model()[2], orc.V, rtol=1e-9, atol=1e-12)


def test_fm_sharded_runner_gpu_engine_and_phase_buffer_alias():
    """carskit_amd.dist.GpuFMEngine: the phase buffer is aliased as a torch tensor (what RCCL all-reduces) and holds
    the same (num, den) the NumPy engine computes; a runner sweep at world 1 equals the fused sweep."""
    import torch
    from carskit_amd import dist as cdist
    from tests.fm_np_engine import NumpyFMEngine
    data = util.small_data(n_users=50, n_items=12, n_dims=2, conds_per_dim=3, n=700, seed=54)
    w0, w, V = fm_init_model(data.n_users, data.n_items, data.n_conds, 4, 2)
    _, a = make_fm(data, 4, 2, DET)
    _, b = make_fm(data, 4, 2, DET)
    a.init()
    b.init()
    ref = NumpyFMEngine(4, data.n_users, data.n_items, data.n_conds, data.n_dims, data.u, data.j, data.ctx, data.r,
                        w0, w, V, REGLW, REGLF, data.n)
    eng = cdist.GpuFMEngine(a, 0)
    for ph in (0, 2, 3, 5):
        a.phase_reduce(ph)
        a.synchronize()
        ref.phase_reduce(ph)
        got = eng.phase_tensor(ph).cpu().numpy()
        want = ref.phase_tensor(ph).numpy()
        np.testing.assert_allclose(got[:1] if ph == 0 else got, want[:1] if ph == 0 else want, rtol=1e-10, atol=1e-12)
    _, c = make_fm(data, 4, 2, DET)
    c.init()
    cdist.ShardedFMRunner(cdist.GpuFMEngine(c, 0), None).sweep()
    c.synchronize()
    b.sweep()
    assert np.array_equal(b.get_model()[2], c.get_model()[2]) and b.get_model()[0] == c.get_model()[0]


def test_fm_phases_in_any_order_match_the_numpy_engine():
    """Errors are never stored (err0 + running delta sums per coordinate), so the phases may be driven in ANY order: drive
    reduce+apply phases in an order the sweep never uses and compare with the dense NumPy phase engine after every step."""
    from tests.fm_np_engine import NumpyFMEngine
    data = util.small_data(n_users=45, n_items=14, n_dims=2, conds_per_dim=3, n=650, seed=55)
    k = 3
    w0, w, V = fm_init_model(data.n_users, data.n_items, data.n_conds, k, 2)
    _, g = make_fm(data, k, 2)
    g.init()
    ref = NumpyFMEngine(k, data.n_users, data.n_items, data.n_conds, data.n_dims, data.u, data.j, data.ctx, data.r,
                        w0, w, V, REGLW, REGLF, data.n)
    order = [2, 3, 2, 5, 6, 5, 9, 3, 0, 8, 4, 12, 11, 10, 6, 6, 1, 0]
    assert max(order) < g.num_phases()
    for ph in order:
        g.phase_reduce(ph)
        ref.phase_reduce(ph)
        g.phase_apply(ph)
        ref.phase_apply(ph)
        gw0, gw, gV = g.get_model()
        np.testing.assert_allclose(gw0, ref.w0, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(gw, ref.w, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(gV, ref.V, rtol=1e-9, atol=1e-12)
    g.sweep()
    for ph in range(ref.num_phases()):
        ref.phase_reduce(ph)
        ref.phase_apply(ph)
    np.testing.assert_allclose(g.get_model()[2], ref.V, rtol=1e-8, atol=1e-11)
    p = g.predict(data.u, data.j, data.ctx)
    assert np.all(np.isfinite(p))


@pytest.mark.parametrize("flags", [DET, RELAXED])
@pytest.mark.parametrize("n_users,n_items,n,zipf", [(3, 40, 1500, None), (1, 30, 900, None), (400, 2, 1200, None),
                                                     (300, 25, 3000, 1.3), (2, 2, 60, None), (1, 40, 6000, None)])
def test_fm_support_length_paths(n_users, n_items, n, zipf, flags):
    """Every reduction path by support length: runs of <= 64 records (one slot, one thread), a hot coordinate's run spread over several
    slots (a COMPLEX coordinate: fm_cplx_kernel adds its slots), a coordinate with more records than one batch holds, coordinates
    without any rating."""
    data = util.small_data(n_users=n_users, n_items=n_items, n_dims=3, conds_per_dim=6 if n >= 6000 else 4, n=n, seed=56, item_zipf=zipf)
    orc, g = make_fm(data, 5, 6, flags)
    if n >= 6000:
        assert np.bincount(data.u).max() > 2048
    orc.init()
    g.init()
    for _ in range(2):
        orc.sweep()
        g.sweep()
    w0, w, V = g.get_model()
    np.testing.assert_allclose(w0, orc.w0, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(w, orc.w, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(V, orc.V.reshape(V.shape), rtol=1e-8, atol=1e-11)


def test_fm_mid_size_sweep_against_the_dense_oracle():
    """The largest problem the DENSE oracle (O(size x p x k) per sweep: FM.java's own cost) finishes in seconds: 40 K ratings,
    p = 2 512, k = 6 -- several groups of coordinates, id-range parts, full batches, complex coordinates with the DEFAULT geometry (no
    test knob), held to the oracle itself rather than to a restatement.  Both forms of the sums."""
    data = util.small_data(n_users=1500, n_items=1000, n_dims=3, conds_per_dim=4, n=40000, seed=77)
    k = 6
    w0, w, V = fm_init_model(data.n_users, data.n_items, data.n_conds, k, 5)
    orc = oracle_c.FMOracle(k, data.n_users, data.n_items, data.n_conds, data.n_dims, data.u, data.j, data.ctx, data.r, w0, w, V, REGLW, REGLF)
    orc.init()
    orc.sweep()
    want = np.array([orc.predict(int(u), int(j), int(c)) for u, j, c in zip(data.u[:4000], data.j[:4000], data.ctx[:4000])])
    for flags in (DET, RELAXED):
        g = capi.FMInstance(k, data.n_users, data.n_items, data.n_conds, data.n_dims, flags=flags)
        g.set_hparams(REGLW, REGLF)
        g.set_ratings(data.u, data.j, data.ctx, data.r)
        g.set_model(w0, w, V)
        g.init()
        g.sweep()
        gw0, gw, gV = g.get_model()
        assert abs(gw0 - orc.w0) <= 1e-10 * max(1.0, abs(orc.w0))
        np.testing.assert_allclose(gw, orc.w, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(gV, orc.V.reshape(gV.shape), rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(g.predict(data.u[:4000], data.j[:4000], data.ctx[:4000]), want, rtol=0, atol=1e-9)
        g.close()


@pytest.mark.parametrize("slice_entries", [8, 16, 1000000])
def test_fm_l2_sliced_orders_match_the_oracle(slice_entries):
    """A block of coordinates walks the SLICES of the gathered table (cells), so that the records a workgroup evaluates together are
    consecutive in gathered-id order inside a small slice and lanes share L2 lines.  Forced here with tiny slices (CMI_FM_SLICE; at
    BASELINE C4's share the default of 32 K entries gives 16 and 20 slices): same model as the dense oracle, whatever the slicing."""
    import os
    data = util.small_data(n_users=60, n_items=40, n_dims=2, conds_per_dim=3, n=3000, seed=58)
    os.environ["CMI_FM_SLICE"] = str(slice_entries)
    try:
        orc, g = make_fm(data, 6, 7)
    finally:
        del os.environ["CMI_FM_SLICE"]
    lay = g.layout()
    if slice_entries < 100:
        assert lay["slices_user_order"] > 1 and lay["slices_item_order"] > 1
    else:
        assert lay["slices_user_order"] == lay["slices_item_order"] == 1
    assert lay["records_user_order"] == lay["records_item_order"] == data.n and lay["records_ctx_order"] == int((data.ctx < data.n_conds).sum())
    orc.init()
    g.init()
    for _ in range(3):
        orc.sweep()
        g.sweep()
    w0, w, V = g.get_model()
    np.testing.assert_allclose(w0, orc.w0, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(w, orc.w, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(V, orc.V.reshape(V.shape), rtol=1e-8, atol=1e-11)
    assert g.time_reduce(4, 2) > 0.0        # the bench's HIP-event timer of one reduce launch: leaves the model alone
    assert np.array_equal(g.get_model()[2], V)


def test_fm_default_form_is_bit_reproducible_and_the_relaxed_form_equals_it_to_rounding():
    """Two trainings on identical inputs give bit-identical models by default, as two runs of the reference's sweep do (FM.java:148-218):
    the default form parks a batch's products in LDS and adds them in an order the layout alone decides.  CMI_FM_FLAG_RELAXED_SUMS adds
    them with LDS atomics as they are evaluated (the order of the fp64 additions varies): same sums to rounding, models within 1e-11
    relative of the default's after three sweeps (both within 1e-8 of the reference arithmetic, tests above).  The round-5 flag
    CMI_FM_FLAG_DETERMINISTIC is still accepted and builds the default form, also next to the relaxed flag or the environment's opt-in."""
    import os
    data = util.small_data(n_users=300, n_items=120, n_dims=2, conds_per_dim=3, n=20000, seed=61)
    runs = []
    for flags, env in ((0, None), (0, None), (capi.FM_FLAG_DETERMINISTIC, None), (capi.FM_FLAG_DETERMINISTIC | RELAXED, None),
                       (capi.FM_FLAG_DETERMINISTIC, "1"), (RELAXED, None), (0, "1")):
        if env: os.environ["CMI_FM_RELAXED_SUMS"] = env
        try:
            _, g = make_fm(data, 8, 3, flags)
        finally:
            os.environ.pop("CMI_FM_RELAXED_SUMS", None)
        g.train(3)          # cmi_fm_train: init + three sweeps
        runs.append(g.get_model())
    for other in runs[1:5]:          # default twice, and every spelling of "deterministic": bit for bit
        assert runs[0][0] == other[0] and np.array_equal(runs[0][1], other[1]) and np.array_equal(runs[0][2], other[2])
    for relaxed in runs[5:]:         # flag and environment opt-in
        np.testing.assert_allclose(relaxed[1], runs[0][1], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(relaxed[2], runs[0][2], rtol=1e-11, atol=1e-13)


def test_fm_runner_exchange_path_on_the_instance_stream_with_rccl():
    """The multi-GPU FM sweep orders  reduce kernel -> all-reduce -> apply kernel  on the instance's own HIP stream (torch's
    ExternalStream), with no host synchronisation.  With one rank the all-reduce is the identity, so the exchanged sweep
    must equal the fused sweep bit for bit -- any missing ordering between the library's kernels and RCCL would show."""
    import os
    import socket
    import torch
    import torch.distributed as tdist
    from carskit_amd import dist as cdist
    data = util.small_data(n_users=300, n_items=120, n_dims=2, conds_per_dim=3, n=20000, seed=57)
    _, a = make_fm(data, 8, 3, DET)
    _, b = make_fm(data, 8, 3, DET)
    a.init()
    b.init()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        eng = cdist.GpuFMEngine(a, 0)
        assert eng.ext is not None
        run = cdist.ShardedFMRunner(eng, tdist, always_exchange=True)
        assert run.lib_comm                        # RCCL: the library issues the per-phase all-reduces itself (cmi_fm_comm_sweep)
        for _ in range(3):
            run.sweep()
            b.sweep()
        a.synchronize()
        torch.cuda.synchronize()
        for x, y in zip(a.get_model(), b.get_model()):
            assert np.array_equal(np.asarray(x), np.asarray(y))
    finally:
        tdist.destroy_process_group()


@pytest.mark.parametrize("batch,slots,slice_entries,shape", [(64, 96, 8, (60, 40, 3000, None)), (32, 96, 4, (300, 25, 3000, 1.3)),
                                                             (128, 96, 16, (1, 40, 6000, None)), (64, 100, 0, (400, 2, 1200, None)),
                                                             (16, 96, 5, (37, 11, 900, None))])
@pytest.mark.parametrize("flags", [DET, RELAXED])
def test_fm_cell_stream_blocks_batches_and_complex_coordinates(batch, slots, slice_entries, shape, flags):
    """The cell stream's structure, forced on small data (CMI_FM_BATCH = records per batch, CMI_FM_SLOTS = accumulator slots per block,
    CMI_FM_SLICE): many blocks, cells cut into several batches, runs longer than 64 records inside a batch (several slots per
    coordinate), coordinates with more records than a block's target (cut over blocks of their own) -- the same model as the dense
    oracle (FM.java:148-218), and the split phases (reduce -> [num | den] -> apply) bit-identical to the fused sweep."""
    import os
    n_users, n_items, n, zipf = shape
    data = util.small_data(n_users=n_users, n_items=n_items, n_dims=3, conds_per_dim=4, n=n, seed=59, item_zipf=zipf)
    os.environ.update(CMI_FM_BATCH=str(batch), CMI_FM_SLOTS=str(slots), CMI_FM_SLICE=str(slice_entries))
    try:
        orc, g = make_fm(data, 5, 8, flags)
        _, g2 = make_fm(data, 5, 8, flags)
    finally:
        for v in ("CMI_FM_BATCH", "CMI_FM_SLOTS", "CMI_FM_SLICE"):
            del os.environ[v]
    lay = g.layout()
    assert lay["batches_user_order"] >= data.n // batch and lay["batches_item_order"] >= data.n // batch
    assert lay["records_user_order"] == lay["records_item_order"] == data.n
    orc.init()
    g.init()
    g2.init()
    for _ in range(2):
        orc.sweep()
        g.sweep()
        for ph in range(g2.num_phases()):
            g2.phase_reduce(ph)
            g2.phase_apply(ph)
    w0, w, V = g.get_model()
    np.testing.assert_allclose(w0, orc.w0, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(w, orc.w, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(V, orc.V.reshape(V.shape), rtol=1e-8, atol=1e-11)
    for x, y in zip(g.get_model(), g2.get_model()):
        if flags == DET:    # the default form: split phases == fused sweep bit for bit
            assert np.array_equal(np.asarray(x), np.asarray(y))
        else:
            np.testing.assert_allclose(np.asarray(x), np.asarray(y), rtol=1e-11, atol=1e-13)

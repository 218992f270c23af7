"""cmi_save_model / cmi_load_model (reference IterativeRecommender.java:249-292): every container survives the file
(the reference forgets condBias / ucBias / icBias), and resuming equals continuous training -- state, losses and bold-driver
rates bit for bit.  A corrupt, truncated or mismatching file is refused and leaves the model untouched."""
import numpy as np
import pytest

from carskit_amd import capi, synth
from tests import util
from tests.test_gpu_parity import make_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,flags", [("BiasedMF", 0), ("PMF", 0), ("CAMF_C", capi.FLAG_SCHED_SERIAL), ("CAMF_CI", 0),
                                         ("CAMF_CU", capi.FLAG_STATE_F64), ("CAMF_CUCI", 0)])
def test_resume_equals_continuous_training(model, flags, tmp_path):
    data = util.small_data(n_users=300, n_items=40, n=4000, seed=61)
    _, a = make_pair(model, data, 64, flags)
    _, b = make_pair(model, data, 64, flags)
    la, ra = a.train(7, util.LR, bold_driver=True)                       # 7 epochs in one go
    lb1, rb1 = b.train(3, util.LR, bold_driver=True)                     # 3 epochs, save ...
    path = tmp_path / "model.cmi"
    b.save_model(path, lrate=b.final_lrate, last_loss=lb1[-1], epochs_done=3)
    saved = b.get_states()
    b.close()
    _, c = make_pair(model, data, 64, flags, seed=99)                    # ... a fresh handle with a DIFFERENT initial model
    c.set_hparams(0.5, 0.5, 0.5, 0.5, 1.0)                               # and wrong hyper-parameters: the file restores both
    lr, last, done = c.load_model(path)
    assert done == 3 and last == lb1[-1]
    for name, arr in c.get_states().items():                            # all containers, context tables included
        assert np.array_equal(arr, saved[name]), name
    lb2, rb2 = c.train(4, lr, bold_driver=True, first_iter=done + 1, prev_loss=last)
    assert np.concatenate([lb1, lb2]).tolist() == la.tolist()
    assert np.concatenate([rb1, rb2]).tolist() == ra.tolist()
    for name, arr in a.get_states().items():
        assert np.array_equal(arr, c.get_states()[name]), name


def test_bad_files_are_refused_and_leave_the_model_alone(tmp_path):
    data = util.small_data(n_users=60, n_items=25, n=900, seed=62)
    _, a = make_pair("CAMF_CI", data, 8, 0)
    a.train_epoch(util.LR)
    good = tmp_path / "good.cmi"
    a.save_model(good)
    raw = good.read_bytes()
    before = a.get_states()
    cases = {"flipped": raw[:200] + bytes([raw[200] ^ 1]) + raw[201:], "truncated": raw[:-100], "not_a_model": b"hello" * 100,
             "future_version": raw[:8] + (99).to_bytes(4, "little") + raw[12:]}
    for name, blob in cases.items():
        p = tmp_path / (name + ".cmi")
        p.write_bytes(blob)
        with pytest.raises(capi.CmiError):
            a.load_model(p)
        for n_, arr in a.get_states().items():
            assert np.array_equal(arr, before[n_]), (name, n_)
    _, other = make_pair("CAMF_CI", data, 16, 0)          # same model, other k: refused
    with pytest.raises(capi.CmiError):
        other.load_model(good)
    with pytest.raises(capi.CmiError):
        a.load_model(tmp_path / "missing.cmi")


@pytest.mark.parametrize("model,sched", [("CAMF_CI", 0), ("CAMF_CU", 0), ("CAMF_CUCI", 0), ("BiasedMF", 0), ("CAMF_C", capi.FLAG_SCHED_SERIAL)])
def test_save_load_then_m_epochs_equals_the_oracles_n_plus_m_epochs(model, sched, tmp_path):
    """The oracle leg of N4: GPU N epochs -> file -> a FRESH handle -> M more epochs must be the ORACLE's
    N + M sequential epochs, strict fp64 bit for bit (model state, and under the serial schedule also every loss and bold-driver
    rate) -- not merely a self round trip."""
    n_ep, m_ep, k = 3, 4, 10
    data = util.small_data(n_users=120, n_items=35, n=2500, seed=63)
    flags = capi.FLAG_STATE_F64 | capi.FLAG_STRICT | sched
    orc, a = make_pair(model, data, k, flags)
    la, ra = a.train(n_ep, util.LR, bold_driver=True)
    path = tmp_path / "m.cmi"
    a.save_model(path, lrate=a.final_lrate, last_loss=la[-1], epochs_done=n_ep)
    a.close()
    _, b = make_pair(model, data, k, flags, seed=77)                     # different initial model: the file must supply everything
    lr, last, done = b.load_model(path)
    lb, rb = b.train(m_ep, lr, bold_driver=True, first_iter=done + 1, prev_loss=last)
    # the oracle: N + M epochs in one go, IterativeRecommender.updateLRate with the bold driver
    rate, lo, rates = util.LR, [], []
    for it in range(1, n_ep + m_ep + 1):
        rates.append(rate)
        lo.append(orc.epoch(rate))
        if it > 1:
            rate = rate * 1.05 if abs(lo[-2]) > abs(lo[-1]) else rate * 0.5
    got_l, got_r = np.concatenate([la, lb]), np.concatenate([ra, rb])
    if sched & capi.FLAG_SCHED_SERIAL:                                   # the reference's running-sum order: identical bits
        assert got_l.tolist() == lo and got_r.tolist() == rates
    else:                                                                # level schedule: same terms, another association of the sum
        np.testing.assert_allclose(got_l, lo, rtol=1e-12)
        assert got_r.tolist() == rates
    for name, arr in b.get_states().items():
        assert np.array_equal(arr, np.asarray(orc.state[name]).reshape(arr.shape)), name


def test_sim_params_travel_with_the_file_and_are_verified(tmp_path):
    """numF / n_ctx_dims / EmptyContextConditions are part of a CAMF_*CS model -- restored into a handle that has none,
    refused when the handle was set up with others."""
    data = util.small_data(n_users=40, n_items=15, n_dims=2, conds_per_dim=3, n=500, seed=64)
    def make(empty):
        inst = capi.Instance("CAMF_LCS", 5, data.n_users, data.n_items, data.n_conds, flags=capi.FLAG_SCHED_SERIAL | capi.FLAG_STATE_F64)
        if empty is not None:
            inst.set_sim_params(4, data.n_dims, empty)
        return inst
    a = make([0, 3])
    a.set_hparams(util.REG, util.REG, util.REG, util.REGC, 3.0)
    a.set_ratings(data.u, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
    a.set_state("cfMatrix", np.random.default_rng(1).random((data.n_conds, 4)))
    a.train_epoch(util.LR)
    p = tmp_path / "lcs.cmi"
    a.save_model(p)
    b = make(None)                                   # no sim params yet: the file restores them, then the ratings can be set
    b.load_model(p)
    b.set_ratings(data.u, data.j, data.ctx, data.r, data.ctx_ptr, data.ctx_conds)
    pa = a.predict(data.u[:20], data.j[:20], data.ctx[:20])
    pb = b.predict(data.u[:20], data.j[:20], data.ctx[:20])
    assert np.array_equal(pa, pb)
    c = make([1, 3])                                 # other EmptyContextConditions: refused
    with pytest.raises(capi.CmiError, match="EmptyContextConditions"):
        c.load_model(p)
    # a handle configured with an EMPTY list of empty conditions (a data set without ':na' columns; the reference's CAMF_*CS
    # cannot train on one -- EmptyContextConditions.get(i) throws -- and neither does cmi_set_ratings, but the handle and its file are
    # legal) restores into a fresh handle too: the restore used to be skipped, and cfMatrix then failed its count check after P and Q
    # had already been overwritten
    d = make([])
    cf = np.random.default_rng(2).random((data.n_conds, 4))
    d.set_state("cfMatrix", cf)
    d.set_state("P", np.full((data.n_users, 5), 0.25))
    q = tmp_path / "lcs_noempty.cmi"
    d.save_model(q)
    e = make(None)
    e.load_model(q)
    assert np.array_equal(e.get_state("cfMatrix"), cf) and np.array_equal(e.get_state("P"), d.get_state("P"))
    with pytest.raises(capi.CmiError, match="EmptyContextConditions"):     # a handle configured with an empty list refuses a file with [0, 3]
        make([]).load_model(p)
    before = e.get_state("P")
    with pytest.raises(capi.CmiError, match="EmptyContextConditions"):     # ... and a refused file leaves the model as it was
        e.load_model(p)
    assert np.array_equal(e.get_state("P"), before)

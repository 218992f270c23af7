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
             "future_version": raw[:8] + (2).to_bytes(4, "little") + raw[12:]}
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

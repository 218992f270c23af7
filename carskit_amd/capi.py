"""ctypes binding of libcarskit_mi355x.so (include/carskit_mi355x.h).

This is the only way Python reaches the compute path; there is no Python or CPU fallback.  If the
shared library is missing (not built) loading raises; without a HIP device cmi_create fails with
CMI_E_NO_DEVICE and `Instance(...)` raises `CmiError`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CMI_LIB_PATH") or os.path.join(_HERE, "lib", "libcarskit_mi355x.so")   # CMI_LIB_PATH: experiments with variant builds

OK, E_INVALID, E_NO_DEVICE, E_HIP, E_NUMERIC, E_UNSUPPORTED, E_HOST, E_BUSY = 0, -1, -2, -3, -4, -5, -6, -7

MODEL_IDS = {"BiasedMF": 0, "CAMF_C": 1, "CAMF_CI": 2, "CAMF_CU": 3, "CAMF_CUCI": 4, "PMF": 5,
             "SVD++": 6, "CAMF_ICS": 7, "CAMF_LCS": 8, "CAMF_MCS": 9}
STATE_IDS = {"P": 0, "Q": 1, "userBias": 2, "itemBias": 3, "condBias": 4, "ucBias": 5, "icBias": 6,
             "Y": 7, "ccMatrix": 8, "cfMatrix": 9, "cVector": 10}
MODEL_STATES = {
    "BiasedMF": ("P", "Q", "userBias", "itemBias"),
    "CAMF_C": ("P", "Q", "userBias", "itemBias", "condBias"),
    "CAMF_CI": ("P", "Q", "userBias", "icBias"),
    "CAMF_CU": ("P", "Q", "itemBias", "ucBias"),
    "CAMF_CUCI": ("P", "Q", "ucBias", "icBias"),
    "PMF": ("P", "Q"),
    "SVD++": ("P", "Q", "userBias", "itemBias", "Y"),
    "CAMF_ICS": ("P", "Q", "ccMatrix"),
    "CAMF_LCS": ("P", "Q", "cfMatrix"),
    "CAMF_MCS": ("P", "Q", "cVector"),
}
# the state keyed by item (replicated and reconciled across GPUs when tuples are sharded by user)
ITEM_SIDE = {"Q", "itemBias", "icBias", "condBias"}

FLAG_STATE_F64 = 0x1
FLAG_SCHED_SERIAL = 0x2
FLAG_STRICT = 0x4
FLAG_NO_GRAPH = 0x10
FLAG_SCHED_CHAIN = 0x80
FLAG_NO_CHAIN = 0x100
FLAG_SCHED_OWNER = 0x200
FLAG_NO_OWNER = 0x400
FLAG_SPOKE_ARENA = 0x800
FLAG_NO_ARENA = 0x1000
FM_FLAG_DETERMINISTIC = 0x1      # round-5 name of what is now the default (accepted, no effect)
FM_FLAG_RELAXED_SUMS = 0x2       # opt into the LDS-atomic sums (faster, last bits vary run to run)
OWN_HUB_FWD, OWN_HUB_LATE, OWN_HUB_STORE, OWN_SPK_FWD, OWN_SPK_STORE = 1, 2, 4, 8, 16

# every symbol include/carskit_mi355x.h declares: (name, restype, argtypes)
_vp, _i64, _i32, _dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
SYMBOLS = [
    ("cmi_abi_version", C.c_int, []),
    ("cmi_device_count", C.c_int, []),
    ("cmi_create", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.POINTER(_vp)]),
    ("cmi_destroy", C.c_int, [_vp]),
    ("cmi_last_error", C.c_char_p, [_vp]),
    ("cmi_set_ratings", C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    ("cmi_set_state", C.c_int, [_vp, C.c_int, _vp, _i64, C.c_int]),
    ("cmi_get_state", C.c_int, [_vp, C.c_int, _vp, _i64, C.c_int]),
    ("cmi_set_sim_params", C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int]),
    ("cmi_set_hparams", C.c_int, [_vp, _dbl, _dbl, _dbl, _dbl, _dbl]),
    ("cmi_set_device_share", C.c_int, [_vp, C.c_int]),
    ("cmi_train_epoch", C.c_int, [_vp, _dbl, C.POINTER(_dbl)]),
    ("cmi_train", C.c_int, [_vp, C.c_int, _dbl, _dbl, C.c_int, _dbl, C.c_int, _vp, _vp, C.POINTER(C.c_int),
                            C.POINTER(_dbl)]),
    ("cmi_train_from", C.c_int, [_vp, C.c_int, _dbl, C.c_int, _dbl, _dbl, C.c_int, _dbl, C.c_int, _vp, _vp, C.POINTER(C.c_int),
                                 C.POINTER(_dbl)]),
    ("cmi_predict_batch", C.c_int, [_vp, _i64, _vp, _vp, _vp, C.c_int, _dbl, _dbl, _vp]),
    ("cmi_eval_ratings", C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _dbl, _dbl, _vp, C.POINTER(_i64)]),
    ("cmi_set_eval_ratings", C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    ("cmi_eval_resident", C.c_int, [_vp, _dbl, _dbl, _vp, C.POINTER(_i64)]),
    ("cmi_eval_rankings", C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _dbl, C.c_int, C.c_int,
                                    C.c_int, _vp, C.POINTER(_i64), _vp, _vp, _vp, _vp, _vp]),
    ("cmi_fm_eval_rankings", C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _dbl, C.c_int, C.c_int,
                                       C.c_int, _vp, C.POINTER(_i64), _vp, _vp, _vp, _vp, _vp]),
    ("cmi_last_rank_ms", C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(_dbl)]),
    ("cmi_last_rank_host_ms", C.c_int, [_vp, _vp]),
    ("cmi_last_rank_kernel_ms", C.c_int, [_vp, _vp]),
    ("cmi_group_set_eval_ratings", C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    ("cmi_group_eval_resident", C.c_int, [_vp, _dbl, _dbl, _vp, C.POINTER(_i64)]),
    ("cmi_comm_unique_id", C.c_int, [_vp]),
    ("cmi_comm_init", C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    ("cmi_comm_exchange", C.c_int, [_vp, _dbl]),
    ("cmi_comm_train_epoch", C.c_int, [_vp, _dbl, _dbl, C.POINTER(_dbl)]),
    ("cmi_comm_last_exchange_ms", C.c_int, [_vp, C.POINTER(C.c_float)]),
    ("cmi_rank_plan", C.c_int, [C.c_int32, C.c_int32, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _dbl, C.c_int,
                                C.POINTER(_i64), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("cmi_rank_list_measures", C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    ("cmi_narrow_runs", C.c_int, [_i64, _vp, _i64, _i64, _vp, C.POINTER(_i64)]),
    ("cmi_conflict_free_blocks", C.c_int, [_i64, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _i64, C.POINTER(_i64)]),
    ("cmi_java_int_hashset_order", C.c_int, [_i64, _vp, _vp, C.POINTER(_i64)]),
    ("cmi_state_device_ptr", C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_i64), C.POINTER(C.c_int)]),
    ("cmi_stream", C.c_int, [_vp, C.POINTER(_vp)]),
    ("cmi_synchronize", C.c_int, [_vp]),
    ("cmi_train_epoch_async", C.c_int, [_vp, _dbl]),
    ("cmi_last_loss", C.c_int, [_vp, C.POINTER(_dbl)]),
    ("cmi_save_model", C.c_int, [_vp, C.c_char_p, C.c_double, C.c_double, C.c_int]),
    ("cmi_load_model", C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    ("cmi_measure_hbm", C.c_int, [C.c_int, _i64, C.POINTER(C.c_double)]),
    ("cmi_schedule_info", C.c_int, [_vp, C.POINTER(_i64)]),
    ("cmi_schedule_traffic", C.c_int, [_vp, C.POINTER(_i64)]),
    ("cmi_schedule_note", C.c_char_p, [_vp]),
    ("cmi_arena_positions", C.c_int, [_i64, _vp, _i32, _vp, _vp]),
    ("cmi_exchange_setup", C.c_int, [_vp, _i64, C.POINTER(_vp), C.POINTER(_i64)]),
    ("cmi_group_create", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_uint, C.POINTER(_vp)]),
    ("cmi_group_destroy", C.c_int, [_vp]),
    ("cmi_group_last_error", C.c_char_p, [_vp]),
    ("cmi_group_size", C.c_int, [_vp]),
    ("cmi_group_set_hparams", C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    ("cmi_group_set_ratings", C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    ("cmi_group_set_state", C.c_int, [_vp, C.c_int, _vp, _i64, C.c_int]),
    ("cmi_group_get_state", C.c_int, [_vp, C.c_int, _vp, _i64, C.c_int]),
    ("cmi_group_train_epoch", C.c_int, [_vp, C.c_double, C.POINTER(C.c_double)]),
    ("cmi_group_set_lr_scale", C.c_int, [_vp, C.c_double]),
    ("cmi_group_train", C.c_int, [_vp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, C.c_int, _vp, _vp, C.POINTER(C.c_int),
                                  C.POINTER(C.c_double)]),
    ("cmi_group_train_from", C.c_int, [_vp, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, C.c_int, _vp, _vp,
                                       C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    ("cmi_group_eval_ratings", C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, C.c_double, C.c_double, _vp, C.POINTER(_i64)]),
    ("cmi_group_predict_batch", C.c_int, [_vp, _i64, _vp, _vp, _vp, C.c_int, C.c_double, C.c_double, _vp]),
    ("cmi_group_shard_info", C.c_int, [_vp, C.c_int, C.POINTER(_i64)]),
    ("cmi_group_last_times", C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("cmi_group_exchange_path", C.c_char_p, [_vp]),
    ("cmi_group_member", C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    ("cmi_exchange_pack", C.c_int, [_vp]),
    ("cmi_exchange_apply", C.c_int, [_vp, C.c_double]),
    ("cmi_loss_device_ptr", C.c_int, [_vp, C.POINTER(_vp)]),
    ("cmi_last_epoch_ms", C.c_int, [_vp, C.POINTER(C.c_float)]),
    ("cmi_level_schedule", C.c_int, [_i64, _vp, _vp, _i32, _i32, C.c_int, _vp, _vp, _i64, C.POINTER(_i64)]),
    ("cmi_owner_schedule", C.c_int, [_i64, _vp, _vp, _i32, _i32, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.POINTER(C.c_int)]),
    ("cmi_chain_schedule", C.c_int, [_i64, _vp, _vp, _i32, _i32, C.c_int, C.c_int, _vp, _vp, _i64, _vp, _i64, C.POINTER(_i64),
                                     C.POINTER(_i64), C.POINTER(C.c_int)]),
    ("cmi_chain_schedule_device", C.c_int, [C.c_int, _i64, _vp, _vp, _i32, _i32, C.c_int, C.c_int, _vp, _vp, _i64, _vp, _i64, C.POINTER(_i64),
                                     C.POINTER(_i64), C.POINTER(C.c_int)]),
    ("cmi_dao_read", C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    ("cmi_dao_read_shared", C.c_int, [C.c_char_p, _vp, C.POINTER(_vp)]),
    ("cmi_dao_destroy", C.c_int, [_vp]),
    ("cmi_dao_last_error", C.c_char_p, [_vp]),
    ("cmi_dao_counts", C.c_int, [_vp, C.POINTER(_i64)]),
    ("cmi_dao_matrix", C.c_int, [_vp, _vp, _vp, _vp]),
    ("cmi_dao_ui_maps", C.c_int, [_vp, _vp, _vp]),
    ("cmi_dao_ctx_nnz", _i64, [_vp]),
    ("cmi_dao_ctx_table", C.c_int, [_vp, _vp, _vp]),
    ("cmi_dao_cond_info", C.c_int, [_vp, _vp, _vp, C.POINTER(_i32)]),
    ("cmi_dao_rating_scale", C.c_int, [_vp, _vp, _i32, C.POINTER(_i32)]),
    ("cmi_dao_raw_id", C.c_char_p, [_vp, C.c_int, _i32]),
    ("cmi_java_hashmap_order", C.c_int, [_i64, _vp, _vp, C.POINTER(C.c_int)]),
    ("cmi_transform_compact_to_binary", C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]),
    ("cmi_validate_data_format", C.c_int, [C.c_char_p]),
    ("cmi_transform", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]),
    ("cmi_fm_create", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.POINTER(_vp)]),
    ("cmi_fm_destroy", C.c_int, [_vp]),
    ("cmi_fm_last_error", C.c_char_p, [_vp]),
    ("cmi_fm_set_hparams", C.c_int, [_vp, _dbl, _dbl, _i64]),
    ("cmi_fm_set_ratings", C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    ("cmi_fm_set_model", C.c_int, [_vp, _dbl, _vp, _vp]),
    ("cmi_fm_get_model", C.c_int, [_vp, C.POINTER(_dbl), _vp, _vp]),
    ("cmi_fm_init", C.c_int, [_vp]),
    ("cmi_fm_sweep", C.c_int, [_vp]),
    ("cmi_fm_train", C.c_int, [_vp, C.c_int]),
    ("cmi_fm_predict_batch", C.c_int, [_vp, _i64, _vp, _vp, _vp, C.c_int, _dbl, _dbl, _vp]),
    ("cmi_fm_synchronize", C.c_int, [_vp]),
    ("cmi_fm_stream", C.c_int, [_vp, C.POINTER(_vp)]),
    ("cmi_fm_num_phases", C.c_int, [_vp]),
    ("cmi_fm_phase_reduce", C.c_int, [_vp, C.c_int]),
    ("cmi_fm_phase_buffer", C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_i64)]),
    ("cmi_fm_phase_apply", C.c_int, [_vp, C.c_int]),
    ("cmi_fm_phase_run", C.c_int, [_vp, C.c_int]),
    ("cmi_fm_layout", C.c_int, [_vp, _vp]),
    ("cmi_fm_comm_init", C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    ("cmi_fm_comm_sweep", C.c_int, [_vp]),
    ("cmi_fm_time_reduce", C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_dbl)]),
]

_LIB = None


RANK_MEASURES = ("Pre5", "Pre10", "PreN", "Rec5", "Rec10", "RecN", "AUC5", "AUC10", "AUCN", "MAP5", "MAP10", "MAPN",
                 "NDCG5", "NDCG10", "NDCGN", "MRR5", "MRR10", "MRRN", "D5", "D10", "DN")


def java_int_hashset_order(values):
    v = np.ascontiguousarray(values, dtype=np.int32)
    out, n = np.zeros(max(len(v), 1), np.int32), _i64()
    rc = lib().cmi_java_int_hashset_order(len(v), _p(v), _p(out), C.byref(n))
    if rc:
        raise CmiError(rc, "cmi_java_int_hashset_order")
    return out[:n.value].copy()


def narrow_runs(level_off, max_tuples=256, min_levels=16):
    off = np.ascontiguousarray(level_off, dtype=np.int64)
    run = np.zeros(max(len(off) - 1, 1), np.int32)
    nl = _i64()
    rc = lib().cmi_narrow_runs(len(off) - 1, _p(off), max_tuples, min_levels, _p(run), C.byref(nl))
    if rc:
        raise CmiError(rc, "cmi_narrow_runs")
    return run[:len(off) - 1].copy(), nl.value


def conflict_free_blocks(u, j, n_users, n_items, max_block=64):
    u = np.ascontiguousarray(u, dtype=np.int32)
    j = np.ascontiguousarray(j, dtype=np.int32)
    nb = _i64()
    rc = lib().cmi_conflict_free_blocks(len(u), _p(u), _p(j), n_users, n_items, max_block, None, 0, C.byref(nb))
    if rc:
        raise CmiError(rc, "cmi_conflict_free_blocks")
    off = np.zeros(nb.value + 1, np.int32)
    rc = lib().cmi_conflict_free_blocks(len(u), _p(u), _p(j), n_users, n_items, max_block, _p(off), len(off), C.byref(nb))
    if rc:
        raise CmiError(rc, "cmi_conflict_free_blocks")
    return off


def rank_list_measures(ranked, truth, num_dropped, num_recs):
    """Host-only: {measure: value} of one ranked list (already cut at num_recs), as cmi_eval_rankings computes per query."""
    rk = np.ascontiguousarray(ranked, dtype=np.int32)
    tr = np.ascontiguousarray(sorted(set(truth)), dtype=np.int32)
    out = np.zeros(18)
    rc = lib().cmi_rank_list_measures(_p(rk), len(rk), _p(tr), len(tr), int(num_dropped), int(num_recs), _p(out))
    if rc:
        raise CmiError(rc, "cmi_rank_list_measures")
    return dict(zip(RANK_MEASURES[:18], out.tolist()))


def rank_plan(n_users, n_items, train, test, bin_thold=-1.0, num_ignore=0):
    """Host-only: (candidates, [(user, ctx, correct items, excluded candidate positions), ...]) as cmi_eval_rankings sees them."""
    c32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    tu, tj, tc, tr = c32(train[0]), c32(train[1]), c32(train[2]), f64(train[3])
    su, sj, sc, sr = c32(test[0]), c32(test[1]), c32(test[2]), f64(test[3])
    sizes = (_i64 * 4)()
    args = (n_users, n_items, len(tu), _p(tu), _p(tj), _p(tc), _p(tr), len(su), _p(su), _p(sj), _p(sc), _p(sr), float(bin_thold),
            int(num_ignore), sizes)
    rc = lib().cmi_rank_plan(*args, None, None, None, None, None, None, None)
    if rc:
        raise CmiError(rc, "cmi_rank_plan")
    nc, nq, nt, ne = list(sizes)
    cand, qu, qc = np.zeros(max(nc, 1), np.int32), np.zeros(max(nq, 1), np.int32), np.zeros(max(nq, 1), np.int32)
    tp, ti = np.zeros(nq + 1, np.int64), np.zeros(max(nt, 1), np.int32)
    ep, ei = np.zeros(nq + 1, np.int64), np.zeros(max(ne, 1), np.int32)
    rc = lib().cmi_rank_plan(*args, _p(cand), _p(qu), _p(qc), _p(tp), _p(ti), _p(ep), _p(ei))
    if rc:
        raise CmiError(rc, "cmi_rank_plan")
    queries = [(int(qu[q]), int(qc[q]), ti[tp[q]:tp[q + 1]].tolist(), ei[ep[q]:ep[q + 1]].tolist()) for q in range(nq)]
    return cand[:nc].tolist(), queries


COMM_ID_BYTES = 128


def comm_unique_id():
    """cmi_comm_unique_id: the RCCL unique id (bytes) rank 0 creates for a one-process-per-GPU job."""
    buf = (C.c_char * COMM_ID_BYTES)()
    rc = lib().cmi_comm_unique_id(buf)
    if rc:
        raise CmiError(rc, "cmi_comm_unique_id")
    return bytes(buf)


class CmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libcarskit_mi355x error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load the C-ABI library (raises OSError if it has not been built)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s not found: build it with `python __graft_entry__.py` or `make -C carskit_amd/csrc` "
                          "(there is no Python/CPU fallback for the compute path)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def device_count():
    return lib().cmi_device_count()


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def measure_hbm(device=0, nbytes=4 << 30):
    """(copy GB/s, random-512-B-row read-modify-write GB/s) measured on the device right now (cmi_measure_hbm)."""
    out = (C.c_double * 2)()
    rc = lib().cmi_measure_hbm(device, nbytes, out)
    if rc != OK:
        raise CmiError(rc, "cmi_measure_hbm")
    return float(out[0]), float(out[1])


def level_schedule(u, j, n_users, n_items, order=0):
    """Host-only: (perm, level_off) of the dependency-level schedule (see cmi_level_schedule)."""
    u = np.ascontiguousarray(u, dtype=np.int32)
    j = np.ascontiguousarray(j, dtype=np.int32)
    n = len(u)
    nl = _i64()
    rc = lib().cmi_level_schedule(n, _p(u), _p(j), n_users, n_items, order, None, None, 0, C.byref(nl))
    if rc != OK:
        raise CmiError(rc, "cmi_level_schedule")
    perm = np.empty(n, dtype=np.int32)
    off = np.empty(nl.value + 1, dtype=np.int64)
    rc = lib().cmi_level_schedule(n, _p(u), _p(j), n_users, n_items, order, _p(perm), _p(off), len(off), C.byref(nl))
    if rc != OK:
        raise CmiError(rc, "cmi_level_schedule")
    return perm, off


def chain_schedule(u, j, n_users, n_items, hub=-1, max_chain=16):
    """Host-only: (perm, unit_off, level_off, hub_is_item) of the hub-chain level schedule (see cmi_chain_schedule)."""
    u = np.ascontiguousarray(u, dtype=np.int32)
    j = np.ascontiguousarray(j, dtype=np.int32)
    n = len(u)
    nu, nl, hub_used = _i64(), _i64(), C.c_int()
    rc = lib().cmi_chain_schedule(n, _p(u), _p(j), n_users, n_items, hub, max_chain, None, None, 0, None, 0, C.byref(nu),
                                  C.byref(nl), C.byref(hub_used))
    if rc != OK:
        raise CmiError(rc, "cmi_chain_schedule")
    perm = np.empty(n, dtype=np.int32)
    unit_off = np.empty(nu.value + 1, dtype=np.int32)
    level_off = np.empty(nl.value + 1, dtype=np.int64)
    rc = lib().cmi_chain_schedule(n, _p(u), _p(j), n_users, n_items, hub, max_chain, _p(perm), _p(unit_off), len(unit_off),
                                  _p(level_off), len(level_off), C.byref(nu), C.byref(nl), C.byref(hub_used))
    if rc != OK:
        raise CmiError(rc, "cmi_chain_schedule")
    return perm, unit_off, level_off, bool(hub_used.value)


def chain_schedule_device(u, j, n_users, n_items, hub=-1, max_chain=16, device=0):
    """The same schedule built on the device (cmi_chain_schedule_device; what cmi_set_ratings uses for large sets)."""
    u = np.ascontiguousarray(u, dtype=np.int32)
    j = np.ascontiguousarray(j, dtype=np.int32)
    n = len(u)
    # (the device build has no count-only form: sized by the host's counts)
    nu, nl, hub_used = _i64(), _i64(), C.c_int()
    rc = lib().cmi_chain_schedule(n, _p(u), _p(j), n_users, n_items, hub, max_chain, None, None, 0, None, 0, C.byref(nu), C.byref(nl), C.byref(hub_used))
    if rc != OK:
        raise CmiError(rc, "cmi_chain_schedule")
    perm = np.empty(n, dtype=np.int32)
    unit_off = np.empty(nu.value + 1, dtype=np.int32)
    level_off = np.empty(nl.value + 1, dtype=np.int64)
    rc = lib().cmi_chain_schedule_device(device, n, _p(u), _p(j), n_users, n_items, hub, max_chain, _p(perm), _p(unit_off), len(unit_off),
                                         _p(level_off), len(level_off), C.byref(nu), C.byref(nl), C.byref(hub_used))
    if rc != OK:
        raise CmiError(rc, "cmi_chain_schedule_device")
    return perm, unit_off[:nu.value + 1], level_off[:nl.value + 1], bool(hub_used.value)


def owner_schedule(u, j, n_users, n_items, n_owners, hub=-1, depth=8):
    """Host-only: (perm, own_off, want, flags, hub_is_item) of the owner (dataflow) schedule (see cmi_owner_schedule)."""
    u = np.ascontiguousarray(u, dtype=np.int32)
    j = np.ascontiguousarray(j, dtype=np.int32)
    n = len(u)
    perm = np.empty(n, dtype=np.int32)
    own_off = np.empty(n_owners + 1, dtype=np.int64)
    want = np.empty(n, dtype=np.uint32)
    flags = np.empty(n, dtype=np.uint32)
    hub_used = C.c_int()
    rc = lib().cmi_owner_schedule(n, _p(u), _p(j), n_users, n_items, hub, n_owners, depth, _p(perm), _p(own_off), _p(want), _p(flags),
                                  C.byref(hub_used))
    if rc != OK:
        raise CmiError(rc, "cmi_owner_schedule")
    return perm, own_off, want, flags, bool(hub_used.value)


def arena_positions(spoke, n_spokes):
    """Host-only: (next, first) of the spoke arena for a stream of spoke row ids (see cmi_arena_positions)."""
    spoke = np.ascontiguousarray(spoke, dtype=np.int32)
    nxt, first = np.empty(len(spoke), dtype=np.int32), np.empty(n_spokes, dtype=np.int32)
    rc = lib().cmi_arena_positions(len(spoke), _p(spoke), n_spokes, _p(nxt), _p(first))
    if rc != OK:
        raise CmiError(rc, "cmi_arena_positions")
    return nxt, first


def _eval_rankings(fn, chk, h, train, test, bin_thold=-1.0, num_recs=10, num_ignore=0, strategy="ucu", with_lists=False):
    """Recommender.evalRankings (Recommender.java:668-964).  train/test: (u, j, ctx, r) array tuples.
    Returns {measure: value}; with_lists also returns {(u, ctx): [(item, score), ...]}."""
    c32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    tu, tj, tc, tr = c32(train[0]), c32(train[1]), c32(train[2]), f64(train[3])
    su, sj, sc, sr = c32(test[0]), c32(test[1]), c32(test[2]), f64(test[3])
    out, nq = np.zeros(len(RANK_MEASURES)), _i64()
    n = max(len(su), 1)
    if with_lists:
        qu, qc, qn = (np.zeros(n, np.int32) for _ in range(3))
        items, scores = np.zeros(n * num_recs, np.int32), np.zeros(n * num_recs)
    else:
        qu = qc = qn = items = scores = None
    chk(fn(h, len(tu), _p(tu), _p(tj), _p(tc), _p(tr), len(su), _p(su), _p(sj), _p(sc),
                                       _p(sr), float(bin_thold), int(num_recs), int(num_ignore),
                                       {"ucu": 0, "uc": 1}[strategy], _p(out), C.byref(nq), _p(qu), _p(qc), _p(qn),
                                       _p(items), _p(scores)))
    res = dict(zip(RANK_MEASURES, out.tolist()))
    res["n_queries"] = nq.value
    if not with_lists:
        return res
    lists = {}
    for q in range(nq.value):
        if qn[q] > 0:
            lists[(int(qu[q]), int(qc[q]))] = [(int(items[q * num_recs + i]), float(scores[q * num_recs + i]))
                                               for i in range(qn[q])]
    return res, lists


class Group:
    """One recommender trained over several GPUs from this one process (a `cmi_group_handle`): ratings sharded by user, item-side
    containers merged (mean of the shards' moves) after every epoch through RCCL, or in-process when shards share a device."""

    def __init__(self, model, k, n_users, n_items, n_conds, n_shards, devices=None, flags=0):
        self.L = lib()
        self.model = model if isinstance(model, str) else {v: n for n, v in MODEL_IDS.items()}[model]
        self.k, self.n_users, self.n_items, self.n_conds, self.num_f = k, n_users, n_items, n_conds, 0
        self.h = _vp()
        dev = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
        rc = self.L.cmi_group_create(MODEL_IDS[self.model], k, n_users, n_items, n_conds, n_shards, _p(dev), flags, C.byref(self.h))
        if rc != OK:
            self.h = None
            raise CmiError(rc, self.L.cmi_group_last_error(None).decode())

    def _chk(self, rc):
        if rc != OK:
            raise CmiError(rc, self.L.cmi_group_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.cmi_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return self.L.cmi_group_size(self.h)

    def set_hparams(self, regU, regI, regB, regC, global_mean):
        self._chk(self.L.cmi_group_set_hparams(self.h, regU, regI, regB, regC, global_mean))

    def set_ratings(self, u, j, ctx, r, ctx_ptr=None, ctx_conds=None):
        c32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        u, j, ctx, ctx_ptr, ctx_conds = c32(u), c32(j), c32(ctx), c32(ctx_ptr), c32(ctx_conds)
        r = np.ascontiguousarray(r, dtype=np.float64)
        n_ctx = 0 if ctx_ptr is None else len(ctx_ptr) - 1
        self._chk(self.L.cmi_group_set_ratings(self.h, len(r), _p(u), _p(j), _p(ctx), _p(r), n_ctx, _p(ctx_ptr), _p(ctx_conds)))

    def set_state(self, name, arr):
        a = np.ascontiguousarray(arr)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64)
        self._chk(self.L.cmi_group_set_state(self.h, STATE_IDS[name], _p(a), a.size, 1 if a.dtype == np.float64 else 0))

    def set_states(self, state):
        for name, arr in state.items():
            if arr is not None:
                self.set_state(name, arr)

    def get_state(self, name, dtype=np.float64):
        out = np.empty(Instance.state_shape(self, name), dtype=dtype)
        self._chk(self.L.cmi_group_get_state(self.h, STATE_IDS[name], _p(out), out.size, 1 if dtype == np.float64 else 0))
        return out

    def get_states(self, dtype=np.float64):
        return {name: self.get_state(name, dtype) for name in MODEL_STATES[self.model]}

    def set_lr_scale(self, scale):
        self._chk(self.L.cmi_group_set_lr_scale(self.h, float(scale)))

    def train_epoch(self, lrate):
        loss = _dbl()
        self._chk(self.L.cmi_group_train_epoch(self.h, lrate, C.byref(loss)))
        return loss.value

    def train(self, num_iters, init_lrate, max_lrate=-1.0, bold_driver=False, decay=-1.0, early_stop=0, first_iter=1, prev_loss=0.0):
        losses, lrs = np.zeros(num_iters), np.zeros(num_iters)
        n, final = C.c_int(0), _dbl(0)
        rc = self.L.cmi_group_train_from(self.h, first_iter, prev_loss, num_iters, init_lrate, max_lrate, int(bold_driver), decay,
                                         early_stop, _p(losses), _p(lrs), C.byref(n), C.byref(final))
        self.iters_run, self.final_lrate = n.value, final.value
        self._chk(rc)
        return losses[:n.value], lrs[:n.value]

    def eval_ratings(self, u, j, ctx, r, min_rate, max_rate):
        c32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        u, j, ctx = c32(u), c32(j), c32(ctx)
        r = np.ascontiguousarray(r, dtype=np.float64)
        out, cnt = np.zeros(5), _i64()
        self._chk(self.L.cmi_group_eval_ratings(self.h, len(r), _p(u), _p(j), _p(ctx), _p(r), min_rate, max_rate, _p(out), C.byref(cnt)))
        res = dict(zip(("MAE", "RMSE", "NMAE", "rMAE", "rRMSE"), out.tolist()))
        res["n"] = cnt.value
        return res

    def set_eval_ratings(self, u, j, ctx, r):
        """Test tuples routed once to the shards that own their users and kept on the devices (`--early-stop MAE|RMSE`)."""
        c32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        u, j, ctx = c32(u), c32(j), c32(ctx)
        r = np.ascontiguousarray(r, dtype=np.float64)
        self._chk(self.L.cmi_group_set_eval_ratings(self.h, len(r), _p(u), _p(j), _p(ctx), _p(r)))

    def eval_resident(self, min_rate, max_rate):
        out, cnt = np.zeros(5), _i64()
        self._chk(self.L.cmi_group_eval_resident(self.h, min_rate, max_rate, _p(out), C.byref(cnt)))
        res = dict(zip(("MAE", "RMSE", "NMAE", "rMAE", "rRMSE"), out.tolist()))
        res["n"] = cnt.value
        return res

    def predict_batch(self, u, j, ctx, bound=False, lo=0.0, hi=0.0):
        c32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        u, j, ctx = c32(u), c32(j), c32(ctx)
        out = np.empty(len(u))
        self._chk(self.L.cmi_group_predict_batch(self.h, len(u), _p(u), _p(j), _p(ctx), int(bound), lo, hi, _p(out)))
        return out

    def shard_info(self, shard):
        info = (_i64 * 6)()
        self._chk(self.L.cmi_group_shard_info(self.h, shard, info))
        return {"user_lo": info[0], "user_hi": info[1], "tuples": info[2], "device": info[3],
                "exchange": ("none", "rccl", "in-process")[info[4]], "bucket_elems": info[5]}

    def exchange_path(self):
        """Which exchange the group runs and why (RCCL after its pre-flight, or the in-process exchange: shared device / fallback)."""
        return self.L.cmi_group_exchange_path(self.h).decode()

    def last_times(self):
        """(compute_ms, exchange_ms) per shard of the most recent epoch (HIP events on the shards' streams)."""
        W = self.size()
        c, x = (C.c_float * W)(), (C.c_float * W)()
        self._chk(self.L.cmi_group_last_times(self.h, c, x))
        return list(c), list(x)

    def member(self, shard):
        """The shard's instance as a borrowed capi.Instance (the group owns it: do not close)."""
        h = _vp()
        self._chk(self.L.cmi_group_member(self.h, shard, C.byref(h)))
        inst = Instance.__new__(Instance)
        inst.L, inst.h, inst.model, inst.k = self.L, h, self.model, self.k
        si = self.shard_info(shard)
        inst.n_users, inst.n_items, inst.n_conds, inst.num_f, inst.flags = si["user_hi"] - si["user_lo"], self.n_items, self.n_conds, 0, 0
        inst.close = lambda: None
        return inst


class Instance:
    """One recommender instance on one GPU (a `cmi_handle`)."""

    def __init__(self, model, k, n_users, n_items, n_conds, device=0, flags=0):
        self.L = lib()
        self.model = model if isinstance(model, str) else {v: n for n, v in MODEL_IDS.items()}[model]
        self.k, self.n_users, self.n_items, self.n_conds = k, n_users, n_items, n_conds
        self.flags = flags
        self.num_f = 0
        self.h = _vp()
        rc = self.L.cmi_create(MODEL_IDS[self.model], k, n_users, n_items, n_conds, device, flags, C.byref(self.h))
        if rc != OK:
            self.h = None
            raise CmiError(rc, self.L.cmi_last_error(None).decode())

    def _chk(self, rc):
        if rc != OK:
            raise CmiError(rc, self.L.cmi_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.cmi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- data ---------------------------------------------------------------------------------------
    def set_sim_params(self, num_f, n_ctx_dims, empty_conds):
        """CAMF_ICS / LCS / MCS: EmptyContextConditions, `-f` of CAMF_LCS, rateDao.numContextDims() (cmi_set_sim_params)"""
        e = np.ascontiguousarray(empty_conds, dtype=np.int32)
        self._chk(self.L.cmi_set_sim_params(self.h, int(num_f), int(n_ctx_dims), _p(e), len(e)))
        self.num_f = int(num_f)

    def set_ratings(self, u, j, ctx, r, ctx_ptr=None, ctx_conds=None):
        c32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        u, j, ctx, ctx_ptr, ctx_conds = c32(u), c32(j), c32(ctx), c32(ctx_ptr), c32(ctx_conds)
        r = np.ascontiguousarray(r, dtype=np.float64)
        n_ctx = 0 if ctx_ptr is None else len(ctx_ptr) - 1
        self._chk(self.L.cmi_set_ratings(self.h, len(r), _p(u), _p(j), _p(ctx), _p(r), n_ctx, _p(ctx_ptr),
                                         _p(ctx_conds)))

    def set_state(self, name, arr):
        a = np.ascontiguousarray(arr)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64)
        self._chk(self.L.cmi_set_state(self.h, STATE_IDS[name], _p(a), a.size, 1 if a.dtype == np.float64 else 0))

    def set_states(self, state):
        for name, arr in state.items():
            if arr is not None:
                self.set_state(name, arr)

    def state_shape(self, name):
        return {"P": (self.n_users, self.k), "Q": (self.n_items, self.k), "userBias": (self.n_users,),
                "itemBias": (self.n_items,), "condBias": (self.n_conds,), "ucBias": (self.n_users, self.n_conds),
                "icBias": (self.n_items, self.n_conds), "Y": (self.n_items, self.k), "ccMatrix": (self.n_conds, self.n_conds),
                "cfMatrix": (self.n_conds, self.num_f), "cVector": (self.n_conds,)}[name]

    def get_state(self, name, dtype=np.float64):
        out = np.empty(self.state_shape(name), dtype=dtype)
        self._chk(self.L.cmi_get_state(self.h, STATE_IDS[name], _p(out), out.size, 1 if dtype == np.float64 else 0))
        return out

    def get_states(self, dtype=np.float64):
        return {name: self.get_state(name, dtype) for name in MODEL_STATES[self.model]}

    def set_hparams(self, regU, regI, regB, regC, global_mean):
        self._chk(self.L.cmi_set_hparams(self.h, regU, regI, regB, regC, global_mean))

    def set_device_share(self, instances):
        """`cv -p on`: how many instances train concurrently on this device (before set_ratings)."""
        self._chk(self.L.cmi_set_device_share(self.h, int(instances)))

    # -- training -----------------------------------------------------------------------------------
    def train_epoch(self, lrate):
        loss = _dbl()
        self._chk(self.L.cmi_train_epoch(self.h, lrate, C.byref(loss)))
        return loss.value

    def train_epoch_async(self, lrate):
        self._chk(self.L.cmi_train_epoch_async(self.h, lrate))

    def last_loss(self):
        loss = _dbl()
        self._chk(self.L.cmi_last_loss(self.h, C.byref(loss)))
        return loss.value

    def train(self, num_iters, init_lrate, max_lrate=-1.0, bold_driver=False, decay=-1.0, early_stop=0, first_iter=1,
              prev_loss=0.0):
        """cmi_train (first_iter == 1) / cmi_train_from: returns (losses, lrates) of the epochs run"""
        losses, lrs = np.zeros(num_iters), np.zeros(num_iters)
        n, final = C.c_int(0), _dbl(0)
        rc = self.L.cmi_train_from(self.h, first_iter, prev_loss, num_iters, init_lrate, max_lrate, int(bold_driver), decay,
                                   early_stop, _p(losses), _p(lrs), C.byref(n), C.byref(final))
        self.iters_run, self.final_lrate = n.value, final.value
        self._chk(rc)
        return losses[:n.value], lrs[:n.value]

    def synchronize(self):
        self._chk(self.L.cmi_synchronize(self.h))

    def stream_ptr(self):
        p = _vp()
        self._chk(self.L.cmi_stream(self.h, C.byref(p)))
        return p.value

    def last_epoch_ms(self):
        ms = C.c_float()
        self._chk(self.L.cmi_last_epoch_ms(self.h, C.byref(ms)))
        return ms.value

    def schedule_info(self):
        info = (_i64 * 8)()
        self._chk(self.L.cmi_schedule_info(self.h, info))
        d = dict(zip(("levels", "max_level", "tuples", "dmax", "state_bytes", "tuple_bytes", "kind", "flow_blocks"),
                     list(info)))
        d["kind"] = ("level", "serial", "flow", "two-lane", "chain-item", "chain-user", "owner-item", "owner-user")[d["kind"]]
        if d["kind"].startswith("owner"):      # owners in the low word, of which teams (three wavefronts each) in the high word
            d["teams"] = d["flow_blocks"] >> 32
            d["flow_blocks"] &= 0xffffffff
        return d

    def schedule_note(self):
        return self.L.cmi_schedule_note(self.h).decode()

    def schedule_traffic(self):
        """HBM bytes per epoch derived from the loaded schedule: {"sector", "own", "algorithmic", "models_reuse"} (cmi_schedule_traffic)"""
        out = (_i64 * 4)()
        self._chk(self.L.cmi_schedule_traffic(self.h, out))
        return {"sector": out[0], "own": out[1], "algorithmic": out[2], "models_reuse": bool(out[3] & 1), "spoke_arena": bool(out[3] & 2)}

    def stream(self):
        s = _vp()
        self._chk(self.L.cmi_stream(self.h, C.byref(s)))
        return s.value

    def save_model(self, path, lrate=0.0, last_loss=0.0, epochs_done=0):
        self._chk(self.L.cmi_save_model(self.h, str(path).encode(), lrate, last_loss, epochs_done))

    def load_model(self, path):
        """-> (lrate, last_loss, epochs_done) stored with the model"""
        lr, ll, ep = C.c_double(), C.c_double(), C.c_int()
        self._chk(self.L.cmi_load_model(self.h, str(path).encode(), C.byref(lr), C.byref(ll), C.byref(ep)))
        if self.model == "CAMF_LCS" and self.n_conds > 0:      # a file may have restored numF into a handle that had none
            self.num_f = self.state_device_ptr("cfMatrix")[1] // self.n_conds
        return lr.value, ll.value, ep.value

    def exchange_setup(self, pad_to=1):
        """(device pointer, element count, numpy dtype) of the item-side exchange bucket; snapshots the current state."""
        ptr, cnt = _vp(), _i64()
        self._chk(self.L.cmi_exchange_setup(self.h, pad_to, C.byref(ptr), C.byref(cnt)))
        return ptr.value, cnt.value, (np.float64 if self.flags & FLAG_STATE_F64 else np.float32)

    def exchange_pack(self):
        self._chk(self.L.cmi_exchange_pack(self.h))

    def exchange_apply(self, scale):
        self._chk(self.L.cmi_exchange_apply(self.h, float(scale)))

    def loss_device_ptr(self):
        p = _vp()
        self._chk(self.L.cmi_loss_device_ptr(self.h, C.byref(p)))
        return p.value

    # -- the library's own exchange for one-process-per-GPU jobs (cmi_comm_*: the function cmi_group_* uses) ------------------
    def comm_init(self, unique_id, rank, world):
        """unique_id: COMM_ID_BYTES bytes from comm_unique_id() on rank 0, handed to every rank by the host's rendezvous."""
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._chk(self.L.cmi_comm_init(self.h, buf, int(rank), int(world)))

    def comm_last_exchange_ms(self):
        ms = C.c_float()
        self._chk(self.L.cmi_comm_last_exchange_ms(self.h, C.byref(ms)))
        return ms.value

    def comm_exchange(self, scale):
        self._chk(self.L.cmi_comm_exchange(self.h, float(scale)))

    def comm_train_epoch(self, lrate, scale):
        loss = _dbl()
        self._chk(self.L.cmi_comm_train_epoch(self.h, float(lrate), float(scale), C.byref(loss)))
        return loss.value

    def state_device_ptr(self, name):
        ptr, cnt, dt = _vp(), _i64(), C.c_int()
        self._chk(self.L.cmi_state_device_ptr(self.h, STATE_IDS[name], C.byref(ptr), C.byref(cnt), C.byref(dt)))
        return ptr.value, cnt.value, (np.float64 if dt.value else np.float32)

    # -- inference ----------------------------------------------------------------------------------
    def predict(self, u, j, ctx=None, bound=None):
        c32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        u, j, ctx = c32(u), c32(j), c32(ctx)
        out = np.empty(len(u))
        lo, hi = bound if bound else (0.0, 0.0)
        self._chk(self.L.cmi_predict_batch(self.h, len(u), _p(u), _p(j), _p(ctx), 1 if bound else 0, lo, hi, _p(out)))
        return out

    def eval_ratings(self, u, j, ctx, r, min_rate, max_rate):
        c32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        u, j, ctx = c32(u), c32(j), c32(ctx)
        r = np.ascontiguousarray(r, dtype=np.float64)
        out, cnt = np.zeros(5), _i64()
        self._chk(self.L.cmi_eval_ratings(self.h, len(r), _p(u), _p(j), _p(ctx), _p(r), min_rate, max_rate, _p(out),
                                          C.byref(cnt)))
        res = dict(zip(("MAE", "RMSE", "NMAE", "rMAE", "rRMSE"), out.tolist()))
        res["n"] = cnt.value
        return res

    def set_eval_ratings(self, u, j, ctx, r):
        """Keep the test tuples on the device for per-epoch evaluation (`--early-stop MAE|RMSE`)."""
        c32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        u, j, ctx = c32(u), c32(j), c32(ctx)
        r = np.ascontiguousarray(r, dtype=np.float64)
        self._chk(self.L.cmi_set_eval_ratings(self.h, len(r), _p(u), _p(j), _p(ctx), _p(r)))

    def eval_resident(self, min_rate, max_rate):
        out, cnt = np.zeros(5), _i64()
        self._chk(self.L.cmi_eval_resident(self.h, min_rate, max_rate, _p(out), C.byref(cnt)))
        res = dict(zip(("MAE", "RMSE", "NMAE", "rMAE", "rRMSE"), out.tolist()))
        res["n"] = cnt.value
        return res

    def last_rank_ms(self):
        ms, fl = C.c_float(), _dbl()
        self._chk(self.L.cmi_last_rank_ms(self.h, C.byref(ms), C.byref(fl)))
        return ms.value, fl.value

    def last_rank_host_ms(self):
        out = np.zeros(5)
        self._chk(self.L.cmi_last_rank_host_ms(self.h, _p(out)))
        return dict(zip(("plan", "setup", "scoring_loop", "tail", "total"), out.tolist()))

    def last_rank_kernel_ms(self):
        out = np.zeros(2)
        self._chk(self.L.cmi_last_rank_kernel_ms(self.h, _p(out)))
        return dict(zip(("contraction", "selection"), out.tolist()))

    def eval_rankings(self, train, test, bin_thold=-1.0, num_recs=10, num_ignore=0, strategy="ucu", with_lists=False):
        """Recommender.evalRankings (Recommender.java:668-964).  train/test: (u, j, ctx, r) array tuples.
        Returns {measure: value}; with_lists also returns {(u, ctx): [(item, score), ...]}."""
        return _eval_rankings(self.L.cmi_eval_rankings, self._chk, self.h, train, test, bin_thold, num_recs, num_ignore,
                              strategy, with_lists)


class FMInstance:
    """The reference's FM recommender on one GPU (a `cmi_fm_handle`)."""

    def __init__(self, k, n_users, n_items, n_conds, n_ctx_dims, device=0, flags=0):
        self.L = lib()
        self.k, self.n_users, self.n_items, self.n_conds = k, n_users, n_items, n_conds
        self.p = n_users + n_items + n_conds
        self.h = _vp()
        rc = self.L.cmi_fm_create(k, n_users, n_items, n_conds, n_ctx_dims, device, flags, C.byref(self.h))
        if rc != OK:
            self.h = None
            raise CmiError(rc, self.L.cmi_fm_last_error(None).decode())

    def _chk(self, rc):
        if rc != OK:
            raise CmiError(rc, self.L.cmi_fm_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.cmi_fm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_hparams(self, regLw, regLf, global_size=0):
        self._chk(self.L.cmi_fm_set_hparams(self.h, regLw, regLf, global_size))

    def set_ratings(self, u, j, ctx, r):
        c32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        u, j, ctx = c32(u), c32(j), c32(ctx)
        r = np.ascontiguousarray(r, dtype=np.float64)
        self._chk(self.L.cmi_fm_set_ratings(self.h, len(r), _p(u), _p(j), _p(ctx), _p(r)))

    def set_model(self, w0, w, V):
        w = np.ascontiguousarray(w, dtype=np.float64).reshape(self.p)
        V = np.ascontiguousarray(V, dtype=np.float64).reshape(self.p, self.k)
        self._chk(self.L.cmi_fm_set_model(self.h, float(w0), _p(w), _p(V)))

    def get_model(self):
        w0, w, V = _dbl(), np.empty(self.p), np.empty((self.p, self.k))
        self._chk(self.L.cmi_fm_get_model(self.h, C.byref(w0), _p(w), _p(V)))
        return w0.value, w, V

    def init(self):
        self._chk(self.L.cmi_fm_init(self.h))

    def sweep(self):
        self._chk(self.L.cmi_fm_sweep(self.h))
        self._chk(self.L.cmi_fm_synchronize(self.h))

    def train(self, num_iters):
        self._chk(self.L.cmi_fm_train(self.h, num_iters))

    def eval_rankings(self, train, test, bin_thold=-1.0, num_recs=10, num_ignore=0, strategy="ucu", with_lists=False):
        """Recommender.evalRankings with FM.predict as the scorer (see Instance.eval_rankings)."""
        return _eval_rankings(self.L.cmi_fm_eval_rankings, self._chk, self.h, train, test, bin_thold, num_recs, num_ignore,
                              strategy, with_lists)

    def num_phases(self):
        return self.L.cmi_fm_num_phases(self.h)

    def phase_reduce(self, phase):
        self._chk(self.L.cmi_fm_phase_reduce(self.h, phase))

    def phase_buffer(self, phase):
        ptr, cnt = _vp(), _i64()
        self._chk(self.L.cmi_fm_phase_buffer(self.h, phase, C.byref(ptr), C.byref(cnt)))
        return ptr.value, cnt.value

    def phase_run(self, phase):
        self._chk(self.L.cmi_fm_phase_run(self.h, phase))

    def phase_apply(self, phase):
        self._chk(self.L.cmi_fm_phase_apply(self.h, phase))

    def synchronize(self):
        self._chk(self.L.cmi_fm_synchronize(self.h))

    def comm_init(self, unique_id, rank, world):
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._chk(self.L.cmi_fm_comm_init(self.h, buf, int(rank), int(world)))

    def comm_sweep(self):
        self._chk(self.L.cmi_fm_comm_sweep(self.h))

    def layout(self):
        out = np.zeros(12, np.int64)
        self._chk(self.L.cmi_fm_layout(self.h, _p(out)))
        keys = ("slices_user_order", "slices_item_order", "records_user_order", "records_item_order", "records_ctx_order",
                "batches_user_order", "batches_item_order", "bytes_per_factor", "bytes_reduce_user", "bytes_reduce_item",
                "slice_entries", "p")
        return dict(zip(keys, out.tolist()))

    def time_reduce(self, phase, reps=10):
        ms = _dbl()
        self._chk(self.L.cmi_fm_time_reduce(self.h, phase, reps, C.byref(ms)))
        return ms.value

    def stream_ptr(self):
        p = _vp()
        self._chk(self.L.cmi_fm_stream(self.h, C.byref(p)))
        return p.value

    def predict(self, u, j, ctx, bound=None):
        c32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        u, j, ctx = c32(u), c32(j), c32(ctx)
        out = np.empty(len(u))
        lo, hi = bound if bound else (0.0, 0.0)
        self._chk(self.L.cmi_fm_predict_batch(self.h, len(u), _p(u), _p(j), _p(ctx), 1 if bound else 0, lo, hi, _p(out)))
        return out

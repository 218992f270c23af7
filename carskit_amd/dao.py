"""Host-side data access mirroring carskit.data.processor.DataDAO / DataTransformer for the binary and compact
rating formats, on top of the C ABI (cmi_dao_*, cmi_transform_*): string/integer work, bit-exact with the
reference's id assignment."""
import ctypes as C

import numpy as np

from . import capi
from .synth import RatingData


class DataDAO:
    """DataDAO.readData over a binary-format file (reference DataDAO.java:166-354)."""

    def __init__(self, path, train=None):
        """train: the training DataDAO whose id maps this (test) DAO shares and extends (`test-set` evaluation)."""
        self.L = capi.lib()
        self.h = C.c_void_p()
        if train is None:
            rc = self.L.cmi_dao_read(str(path).encode(), C.byref(self.h))
        else:
            rc = self.L.cmi_dao_read_shared(str(path).encode(), train.h, C.byref(self.h))
        if rc != capi.OK:
            self.h = None
            raise capi.CmiError(rc, self.L.cmi_dao_last_error(None).decode())
        cnt = (C.c_int64 * 8)()
        self.L.cmi_dao_counts(self.h, cnt)
        (self.num_users, self.num_items, self.num_user_items, self.num_contexts, self.num_conditions,
         self.num_context_dims, self.num_ratings, self.nnz) = list(cnt)
        n = self.nnz
        self.ui = np.empty(n, np.int32)
        self.ctx = np.empty(n, np.int32)
        self.r = np.empty(n, np.float64)
        self.L.cmi_dao_matrix(self.h, capi._p(self.ui), capi._p(self.ctx), capi._p(self.r))
        self.ui_user = np.empty(self.num_user_items, np.int32)
        self.ui_item = np.empty(self.num_user_items, np.int32)
        self.L.cmi_dao_ui_maps(self.h, capi._p(self.ui_user), capi._p(self.ui_item))
        self.ctx_ptr = np.empty(self.num_contexts + 1, np.int32)
        self.ctx_conds = np.empty(self.L.cmi_dao_ctx_nnz(self.h), np.int32)
        self.L.cmi_dao_ctx_table(self.h, capi._p(self.ctx_ptr), capi._p(self.ctx_conds))
        self.cond_dim = np.empty(self.num_conditions, np.int32)
        empty = np.empty(max(1, self.num_conditions), np.int32)
        ne = C.c_int32()
        self.L.cmi_dao_cond_info(self.h, capi._p(self.cond_dim), capi._p(empty), C.byref(ne))
        self.empty_context_conditions = empty[:ne.value].tolist()
        ns = C.c_int32()
        self.L.cmi_dao_rating_scale(self.h, None, 0, C.byref(ns))
        scale = np.empty(ns.value)
        self.L.cmi_dao_rating_scale(self.h, capi._p(scale), ns.value, C.byref(ns))
        self.rating_scale = scale.tolist()

    def raw(self, kind, idx):
        k = {"user": 0, "item": 1, "cond": 2, "ctx": 3, "dim": 4, "ui": 5}[kind]
        s = self.L.cmi_dao_raw_id(self.h, k, idx)
        return None if s is None else s.decode()

    def raw_ids(self, kind):
        n = {"user": self.num_users, "item": self.num_items, "cond": self.num_conditions, "ctx": self.num_contexts,
             "dim": self.num_context_dims, "ui": self.num_user_items}[kind]
        return [self.raw(kind, i) for i in range(n)]

    def rating_data(self):
        """The matrix as the tuple arrays the recommenders take (u, j per entry through the ui maps)."""
        return RatingData(self.num_users, self.num_items, self.num_conditions, self.num_context_dims,
                          self.ui_user[self.ui], self.ui_item[self.ui], self.ctx.copy(), self.r.copy(), self.ctx_ptr,
                          self.ctx_conds, self.rating_scale[0], self.rating_scale[-1], {"source": "DataDAO"},
                          np.asarray(self.empty_context_conditions, dtype=np.int32))

    def close(self):
        if getattr(self, "h", None):
            self.L.cmi_dao_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def java_hashmap_order(keys):
    """Iteration order (indices into `keys`, which must be distinct) of a default java.util.HashMap<String,?>."""
    arr = (C.c_char_p * len(keys))(*[k.encode("latin-1") for k in keys])
    pos = np.empty(len(keys), np.int64)
    tree = C.c_int()
    rc = capi.lib().cmi_java_hashmap_order(len(keys), arr, capi._p(pos), C.byref(tree))
    if rc != capi.OK:
        raise capi.CmiError(rc, "cmi_java_hashmap_order")
    return pos.tolist(), bool(tree.value)


def transform_compact_to_binary(in_path, out_path):
    tree = C.c_int()
    rc = capi.lib().cmi_transform_compact_to_binary(str(in_path).encode(), str(out_path).encode(), C.byref(tree))
    if rc != capi.OK:
        raise capi.CmiError(rc, capi.lib().cmi_dao_last_error(None).decode())
    return bool(tree.value)


def validate_data_format(path):
    """1 binary, 2 loose, 3 compact (CARSKit.validateDataFormat)."""
    return capi.lib().cmi_validate_data_format(str(path).encode())


def transform(train_in, train_out, test_in=None, test_out=None):
    """DataTransformer.run(): rewrite the rating file(s) in the binary format; returns True if a HashMap bin reached the
    treeify threshold (row order then not guaranteed to be the reference's)."""
    tree = C.c_int()
    rc = capi.lib().cmi_transform(str(train_in).encode(), str(train_out).encode(),
                                  None if test_in is None else str(test_in).encode(),
                                  None if test_out is None else str(test_out).encode(), C.byref(tree))
    if rc != capi.OK:
        raise capi.CmiError(rc, capi.lib().cmi_dao_last_error(None).decode())
    return bool(tree.value)

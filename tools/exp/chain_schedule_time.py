"""Host-only timing + digest of cmi_chain_schedule (the hub-chain level schedule cmi_set_ratings builds) on a C3-like tuple set; no GPU."""
import sys, time, os, hashlib, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from carskit_amd import capi, synth
from carskit_amd.capi import _p, _i64

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
nu_, ni_ = (n // 50, n // 500) if len(sys.argv) < 4 else (int(sys.argv[2]), int(sys.argv[3]))
data = synth.generate_fast(nu_, ni_, 4, 8, n)
u, j = np.ascontiguousarray(data.u, np.int32), np.ascontiguousarray(data.j, np.int32)
L = capi.lib()
for hub in (-3, 1):
    nu, nl, hub_used = _i64(), _i64(), C.c_int()
    t0 = time.perf_counter()
    assert L.cmi_chain_schedule(n, _p(u), _p(j), data.n_users, data.n_items, hub, 16, None, None, 0, None, 0, C.byref(nu), C.byref(nl), C.byref(hub_used)) == 0
    t1 = time.perf_counter()
    perm, unit_off, level_off = np.empty(n, np.int32), np.empty(nu.value + 1, np.int32), np.empty(nl.value + 1, np.int64)
    assert L.cmi_chain_schedule(n, _p(u), _p(j), data.n_users, data.n_items, hub, 16, _p(perm), _p(unit_off), len(unit_off), _p(level_off), len(level_off),
                                C.byref(nu), C.byref(nl), C.byref(hub_used)) == 0
    t2 = time.perf_counter()
    h = hashlib.sha1()
    for a in (perm, unit_off, level_off):
        h.update(a.tobytes())
    print("hub %d -> item=%d: sizes-only call %.2f s, full call %.2f s; units %d levels %d; digest %s" %
          (hub, hub_used.value, t1 - t0, t2 - t1, nu.value, nl.value, h.hexdigest()[:16]))

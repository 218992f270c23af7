"""usage (GPU box): CMI_SETUP_TIMES=1 python tools/exp/device_schedule_time.py users items ratings [hub]  -- wall time of the device-built hub-chain
schedule (cmi_chain_schedule_device) on a synthetic set of that shape, with its phases on stderr; prints a digest to compare with
tools/exp/chain_schedule_time.py (the host builder)."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from carskit_amd import capi, synth
nu, ni, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
hub = int(sys.argv[4]) if len(sys.argv) > 4 else -3
d = synth.generate_fast(nu, ni, 4, 8, n)
u, j = np.ascontiguousarray(d.u, dtype=np.int32), np.ascontiguousarray(d.j, dtype=np.int32)
L = capi.lib()
import ctypes as C
perm = np.empty(n, np.int32); unit_off = np.empty(n + 1, np.int32); level_off = np.empty(1 << 20, np.int64)
nun, nl, hu = C.c_int64(), C.c_int64(), C.c_int()
t0 = time.perf_counter()
rc = L.cmi_chain_schedule_device(0, n, capi._p(u), capi._p(j), d.n_users, d.n_items, hub, 16, capi._p(perm), capi._p(unit_off), len(unit_off), capi._p(level_off), len(level_off), C.byref(nun), C.byref(nl), C.byref(hu))
dt = time.perf_counter() - t0
h = hashlib.sha256(); h.update(perm.tobytes()); h.update(unit_off[:nun.value + 1].tobytes()); h.update(level_off[:nl.value + 1].tobytes())
print("device schedule %d x %d x %d hub %d: rc %d, %.3f s, %d units, %d levels, hub_item %d, digest %s" % (nu, ni, n, hub, rc, dt, nun.value, nl.value, hu.value, h.hexdigest()[:16]))

"""Host-only: writes a synthetic binary-format rating file (3 M lines, then copies with other user / item prefixes up to the requested
size) and times cmi_dao_read on it, sequential and ranged; no GPU.  usage: dao_read_time.py [million lines, multiple of 3]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from carskit_amd import dao

mult = max(1, (int(sys.argv[1]) if len(sys.argv) > 1 else 3) // 3)
base, big = "/tmp/dao_base.csv", "/tmp/dao_big.csv"
n = 3_000_000
rng = np.random.default_rng(1)
u, i, r = rng.integers(0, 200_000, n), rng.integers(0, 20_000, n), rng.integers(1, 6, n)
dims = [("time", 4), ("loc", 4), ("comp", 4), ("mood", 4)]
hdr = "User, Item, Rating, " + ", ".join("%s:%d" % (d, c) for d, k in dims for c in range(k))
flags = []
for d, k in dims:
    c = rng.integers(0, k, n)
    flags.append(np.array([",".join("1" if x == y else "0" for y in range(k)) for x in range(k)])[c])
t0 = time.time()
with open(base, "w") as f:
    f.write("\n".join("u%d,i%d,%d,%s,%s,%s,%s" % t for t in zip(u.tolist(), i.tolist(), r.tolist(), *[a.tolist() for a in flags])) + "\n")
with open(big, "w") as f:
    f.write(hdr + "\n")
for m in range(mult):
    subprocess.check_call("sed 's/^u/%s/;s/,i/,%s/' %s >> %s" % ("uvwxyzabcdefgh"[m], "ijklmnopqrstuv"[m], base, big), shell=True)
print("wrote %d lines, %.0f MB in %.0f s" % (n * mult, os.path.getsize(big) / 1e6, time.time() - t0), flush=True)
for label, env in (("ranged", {}), ("sequential", {"CMI_DAO_PARALLEL_MIN_LINES": "100000000000"})):
    os.environ.pop("CMI_DAO_PARALLEL_MIN_LINES", None)
    os.environ.update(env)
    L = dao.capi.lib()
    import ctypes as C
    best = 1e9
    for _ in range(2):
        h = C.c_void_p()
        t0 = time.perf_counter()
        assert L.cmi_dao_read(big.encode(), C.byref(h)) == 0
        best = min(best, time.perf_counter() - t0)
        cnt = (C.c_int64 * 8)()
        L.cmi_dao_counts(h, cnt)
        L.cmi_dao_destroy(h)
    print("%s: cmi_dao_read %.2f s = %.0f ns per line; counts %s" % (label, best, best / (n * mult) * 1e9, list(cnt)), flush=True)

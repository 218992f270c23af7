"""usage (GPU box): python tools/exp/hipmalloc_probe.py [GB]  -- wall time of one large hipMalloc at several points of a process's life
(start, after a free, after other allocations, beside a running kernel): what cmi_set_ratings' spoke-arena allocation pays and when."""
import ctypes as C, os, sys, time
gb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
hip = C.CDLL("libamdhip64.so")
def malloc(nbytes):
    p = C.c_void_p()
    t = time.perf_counter()
    rc = hip.hipMalloc(C.byref(p), C.c_size_t(nbytes))
    return p, rc, time.perf_counter() - t
def free(p):
    t = time.perf_counter()
    hip.hipFree(p)
    return time.perf_counter() - t
hip.hipSetDevice(0)
for label in ("first call of the process", "after freeing it", "again"):
    p, rc, dt = malloc(gb << 30)
    print("%-40s hipMalloc(%d GB) rc %d %.3f s; hipFree %.3f s" % (label, gb, rc, dt, free(p)))
small = [malloc(1 << 30) for _ in range(8)]
print("8 x 1 GB: %s s" % " ".join("%.3f" % s[2] for s in small))
p, rc, dt = malloc(gb << 30)
print("%-40s hipMalloc(%d GB) rc %d %.3f s" % ("with 8 GB held", gb, rc, dt))
free(p)
for s in small:
    free(s[0])
hip.hipDeviceSynchronize()
time.sleep(3.0)
p, rc, dt = malloc(gb << 30)
print("%-40s hipMalloc(%d GB) rc %d %.3f s" % ("3 s after everything was freed", gb, rc, dt))
free(p)

"""hipMalloc time by size under the HIP runtime torch ships vs the system one (run each mode in its own process)."""
import ctypes as C
import sys
import time

mode = sys.argv[1]
if mode == "torch":
    import torch
    torch.zeros(1, device="cuda")
    import os
    libs = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln]
    print("loaded:", sorted(set(libs)))
    hip = C.CDLL(sorted(set(libs))[0])
else:
    hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
    print("loaded: /opt/rocm/lib/libamdhip64.so")
hip.hipSetDevice(0)
hip.hipFree(None)
v = C.c_int()
hip.hipRuntimeGetVersion(C.byref(v))
print("runtime version", v.value)
for gb in (1, 4, 8):
    p = C.c_void_p()
    t = time.perf_counter()
    rc = hip.hipMalloc(C.byref(p), C.c_size_t(gb << 30))
    t1 = time.perf_counter()
    hip.hipFree(p)
    print("hipMalloc %d GiB rc=%d: %.1f ms; free %.1f ms" % (gb, rc, (t1 - t) * 1e3, (time.perf_counter() - t1) * 1e3))

#!/usr/bin/env python3
"""Probe: does the first open() of a freshly written file hold up reads of OTHER files by other threads?  (The 125 opens of a
hot_path_batch job cost 0.3 s whether issued up front or by a thread beside the readers.)"""
import os, sys, time, tempfile, shutil, threading
base = sys.argv[1] if len(sys.argv) > 1 else None
tmp = tempfile.mkdtemp(prefix="openstall_", dir=base)
N, SIZE = (int(os.environ.get("PROBE_N", "40"))), 432 << 20
try:
    if "--hip" in sys.argv:
        import torch
        torch.zeros(1, device="cuda"); torch.cuda.synchronize()
    blob = os.urandom(1 << 20) * (SIZE >> 20)
    paths = [os.path.join(tmp, "f%02d" % i) for i in range(N)]

    def write(p):
        with open(p, "wb") as f:
            f.write(blob)
    if "--threads" in sys.argv:
        import concurrent.futures
        with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
            list(ex.map(write, paths))
    else:
        for p in paths:
            write(p)
    if "--sync" in sys.argv:
        t0 = time.perf_counter(); os.sync(); print("sync %.2f s" % (time.perf_counter() - t0))
    t0 = time.perf_counter(); warm = [os.open(p, os.O_RDONLY) for p in paths[:8]]
    print("first 8 opens: %.2f ms each" % ((time.perf_counter() - t0) / 8 * 1e3))
    stop = threading.Event()
    log = []                      # (time, bytes) per pread

    def reader(fd):
        buf = bytearray(16 << 20)
        off = 0
        while not stop.is_set():
            n = os.preadv(fd, [buf], off)
            log.append((time.perf_counter(), n))
            off = (off + n) % (SIZE - (16 << 20))
    ths = [threading.Thread(target=reader, args=(fd,)) for fd in warm[:4]]
    t_start = time.perf_counter()
    for t in ths: t.start()
    time.sleep(0.3)
    t_open0 = time.perf_counter()
    per = []
    for p in paths[8:]:
        t0 = time.perf_counter(); fd = os.open(p, os.O_RDONLY); per.append(time.perf_counter() - t0); os.close(fd)
    t_open1 = time.perf_counter()
    time.sleep(0.3)
    stop.set()
    for t in ths: t.join()
    def rate(a, b):
        return sum(n for (t, n) in log if a <= t < b) / max(b - a, 1e-9) / 1e9
    print("opens: %d in %.3f s (%.2f ms each, max %.2f ms)" % (len(per), t_open1 - t_open0, sum(per) / len(per) * 1e3, max(per) * 1e3))
    print("4 readers: %.1f GB/s before, %.1f GB/s during the opens, %.1f GB/s after" % (rate(t_start + 0.05, t_open0), rate(t_open0, t_open1), rate(t_open1, t_open1 + 0.3)))
finally:
    shutil.rmtree(tmp, ignore_errors=True)

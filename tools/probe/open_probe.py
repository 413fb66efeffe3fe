#!/usr/bin/env python3
"""Probe: what the first stat / open / posix_fadvise of a freshly written file costs (the ingest's prepare step took 0.27 s
for 125 files the first time and 0.6 ms the second)."""
import os, sys, time, tempfile, shutil
tmp = tempfile.mkdtemp(prefix="openprobe_", dir=sys.argv[1] if len(sys.argv) > 1 else None)
try:
    blob = os.urandom(1 << 20) * 256
    paths = []
    for i in range(24):
        d = os.path.join(tmp, "s%02d" % i); os.makedirs(d)
        p = os.path.join(d, "reads.all.pileup"); paths.append(p)
        with open(p, "wb") as f:
            f.write(blob)
    for label, fn in (("stat", lambda p: os.stat(p)), ("open", lambda p: os.open(p, os.O_RDONLY)),
                      ("open again", lambda p: os.open(p, os.O_RDONLY))):
        t0 = time.perf_counter(); out = [fn(p) for p in paths[:12]]; dt = time.perf_counter() - t0
        print("%-12s %.2f ms per file" % (label, dt / 12 * 1e3))
        fds = out if label.startswith("open") else []
        if label == "open":
            t0 = time.perf_counter()
            for fd in fds: os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_SEQUENTIAL)
            print("%-12s %.2f ms per file" % ("fadvise", (time.perf_counter() - t0) / 12 * 1e3))
        for fd in fds: os.close(fd)
    t0 = time.perf_counter(); fds = [os.open(p, os.O_RDONLY) for p in paths[12:]]; print("open w/o stat %.2f ms per file" % ((time.perf_counter() - t0) / 12 * 1e3))
    t0 = time.perf_counter(); [os.fstat(fd) for fd in fds]; print("fstat        %.2f ms per file" % ((time.perf_counter() - t0) / 12 * 1e3))
finally:
    shutil.rmtree(tmp, ignore_errors=True)

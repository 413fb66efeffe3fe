"""SNPGPU_TIMING=1: where the wall time of a CLI process goes (interpreter + imports, library load, device context, inputs,
site set, streamed call, output files), printed to stderr when the subcommand finishes.  Off by default and free then."""
import os
import sys
import time

ENABLED = os.environ.get("SNPGPU_TIMING") == "1"
_marks = []
_t0 = time.perf_counter()


def process_start_seconds_ago():
    """Seconds since the process was started (its interpreter start-up and imports are part of a per-sample CLI call)."""
    try:
        with open("/proc/self/stat") as f:
            fields = f.read().rsplit(")", 1)[1].split()
        start_ticks = int(fields[19])
        with open("/proc/uptime") as f:
            uptime = float(f.read().split()[0])
        return uptime - start_ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, IndexError):
        return None


def mark(label):
    if ENABLED:
        _marks.append((label, time.perf_counter()))


def report(what=""):
    if not ENABLED:
        return
    since = process_start_seconds_ago()
    out = ["# SNPGPU_TIMING %s" % what]
    prev = _t0
    if since is not None:
        out.append("#   %-28s %8.1f ms" % ("process start -> imports done", (since - (time.perf_counter() - _t0)) * 1e3))
    for label, t in _marks:
        out.append("#   %-28s %8.1f ms" % (label, (t - prev) * 1e3))
        prev = t
    out.append("#   %-28s %8.1f ms" % ("total since process start" if since is not None else "total", ((since if since is not None else time.perf_counter() - _t0)) * 1e3))
    sys.stderr.write("\n".join(out) + "\n")

"""bench.py's compact line, without a GPU: the extract of a full-size result (the round-5 line, 21 KB, which the driver could not
parse) stays under the limit and keeps what the driver's record needs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _full_result():
    with open(os.path.join(ROOT, "profiles", "r5", "bench_n1.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_compact_line_of_a_full_result_is_short_and_complete():
    import bench
    full = _full_result()
    assert len(json.dumps(full)) > 15000
    line = json.dumps(bench.compact(full), separators=(",", ":"))
    assert len(line.encode()) < bench.COMPACT_LIMIT * 3 // 4
    c = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert c[key] == full[key] or abs(c[key] - full[key]) <= 1e-5 * abs(full[key]), key
    assert c["config"]["workload"] == full["config"]["workload"] and "model" not in c["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in c["roofline"], key
    assert abs(c["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-5 and c["roofline"]["launches"] == full["roofline"]["launches"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c["cpu_baseline"], key
    assert c["cpu_baseline"]["parallel"]["processes"] == full["cpu_baseline"]["parallel"]["processes"]
    assert abs(c["k2_frac"] - full["call_variants"]["strict"]["roofline"]["frac"]) < 1e-5
    assert abs(c["site_calling_frac"] - full["site_calling"]["roofline"]["frac"]) < 1e-5
    assert set(c["scan_shapes_frac"]) == set(full["scan_shapes"])


def test_minimal_line_when_the_extract_would_be_too_long():
    import bench
    full = _full_result()
    full["config"]["workload"] = "w" * 3000
    assert len(json.dumps(bench.compact(full), separators=(",", ":"))) >= bench.COMPACT_LIMIT
    c = bench.compact(full, minimal=True)
    assert "roofline" in c and "cpu_baseline" in c and "secondary" not in c


def test_live_traffic_reads_the_counter_files_of_its_child_runs(tmp_path, monkeypatch):
    """bench_rows.live_traffic without a GPU: a stand-in `rocprofv3` on PATH writes the counter_collection.csv a --pmc pass leaves (four
    launches of the scan kernel among other kernels) — FETCH_SIZE counts 64 bytes per 128-byte line, so it is doubled; WRITE_SIZE is
    taken as it is; KB = 1024 bytes — and the ways a pass can fail end in {"error": ...}, never in an exception."""
    import argparse
    import stat
    import bench_rows
    fake = tmp_path / "bin"
    fake.mkdir()
    script = fake / "rocprofv3"
    script.write_text("""#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
counter, out = a[a.index("--pmc") + 1], a[a.index("-d") + 1]
mode = os.environ.get("FAKE_ROCPROF", "ok")
if mode == "crash":
    sys.stderr.write("rocprofv3: no such counter\\n"); sys.exit(7)
os.makedirs(os.path.join(out, "host", "123"), exist_ok=True)
with open(os.path.join(out, "host", "123", "123_counter_collection.csv"), "w") as f:
    f.write('"Correlation_Id","Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\\n')
    value = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 10.0}[counter]
    if mode != "no_scan":
        for k in range(4):
            f.write('%d,%d,"void k_scan_wave<false, 0>(ScanArgs, SiteSetDev)","%s",%r\\n' % (k, k, counter, value + k))
    f.write('9,9,"void k_scan_wave<true, 0>(ScanArgs, SiteSetDev)","%s",5.0\\n' % counter)
    f.write('10,10,"k_call_mode(SampleDev const*, unsigned int, unsigned int*)","%s",77.0\\n' % counter)
assert "--no-live-traffic" in a and a[a.index("--") + 2].endswith("bench.py")
""")
    script.chmod(script.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(fake) + os.pathsep + os.environ["PATH"])
    args = argparse.Namespace(samples=4, genome=60000, sites=600, depth=30.0, vcf_records=60)
    got = bench_rows.live_traffic(args, 1_000_000.0)
    fetch, write = (1000.0 + 1.5) * 1024 * 2, (10.0 + 1.5) * 1024
    assert got["traffic"] == fetch + write and got["traffic_fetch_bytes"] == fetch and got["traffic_write_bytes"] == write
    assert got["traffic_launches_measured"] == 4 and got["traffic_over_algorithmic"] == (fetch + write) / 1e6
    assert got["traffic_source"].startswith("measured in this run")
    monkeypatch.setenv("FAKE_ROCPROF", "crash")
    assert "exit code 7" in bench_rows.live_traffic(args, 1e6)["error"]
    monkeypatch.setenv("FAKE_ROCPROF", "no_scan")
    assert "0 launches seen" in bench_rows.live_traffic(args, 1e6)["error"]

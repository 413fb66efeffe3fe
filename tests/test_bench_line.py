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

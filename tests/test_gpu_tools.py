"""The timing helpers of tools/ at toy sizes: they are the lab notebook behind DESIGN.md's numbers and had no test (VERDICT r3 weak
#9); a helper that no longer runs against the current library is found here, not in the next tuning session."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("argv, expect", [
    (["scan_tune.py", "3", "60000", "batch", "20"], "scan "),
    (["scan_tune.py", "2", "60000", "single", "20"], "scan "),
    (["varscan_kernel_time.py", "200000", "30", "2"], "records:"),
    (["varscan_kernel_time.py", "200000", "30", "2", "3"], "batch of 3 samples"),
    (["varscan_time.py", "200000", "30", "1"], ""),
    (["varscan_files_time.py", "3", "1"], ""),
    (["pipeline_time.py", "--samples", "4", "--genome", "60000", "--sites", "600", "--separate", "--resident-frac", "0.5"], "outputs_identical_to_fully_resident\": true"),
    (["host_steps_time.py", "30", "2000"], ""),
    (["merge_scale.py", "50", "60", "2000"], ""),
    (["distance_cli_time.py", "60", "900"], ""),
    (["scan_multi_contig.py", "5", "20000", "2"], ""),
    (["scan_crlf.py"], ""),
    (["scan_sweep.py", "3", "20", "", "OVERSUB=1"], "GB/s"),
    (["scan_realloc.py", "3", "20", "2"], "round 1:"),
    (["scan_placement.py", "20", "3"], "ctx 0, input 6"),
    (["scan_batch_sizes.py", "20", "5"], "all again"),
])
def test_tool_runs_at_toy_size(argv, expect, tmp_path):
    env = dict(os.environ, TMPDIR=str(tmp_path), SWEEP_GENOME="60000")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", argv[0])] + argv[1:], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (argv, r.stdout[-1500:], r.stderr[-2500:])
    assert expect in r.stdout, (argv, r.stdout[-1500:])

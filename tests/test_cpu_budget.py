"""The host side's CPU budget (csrc/host_budget.hip; run.py:387-400 is the reference's MaxCpuCores cap): the library sizes its reader
and writer pools from the affinity mask, the cgroup quota and the number of ranks that share the node — not from the CPU count of
the box (256 on the bench box, of which the cgroup grants 16).  No GPU needed: snpgpu_cpu_budget touches no device."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = """
import json, os, sys
sys.path.insert(0, %r)
keep = sorted(os.sched_getaffinity(0))[:int(os.environ["PROBE_AFFINITY"])]
os.sched_setaffinity(0, keep)
from snp_pipeline_amd import device as dev
out = {"first": dev.cpu_budget()}
if os.environ.get("PROBE_SET_RANKS"):
    dev.set_local_ranks(int(os.environ["PROBE_SET_RANKS"]))
    out["after_set_ranks"] = dev.cpu_budget()
    dev.set_local_ranks(0)
if os.environ.get("PROBE_SET_CORES"):
    dev.set_max_cpu_cores(int(os.environ["PROBE_SET_CORES"]))
    out["after_set_cores"] = dev.cpu_budget()
print(json.dumps(out))
""" % ROOT


def _probe(tmp_path, affinity, cpu_max=None, v1=None, **env):
    root = tmp_path / "cgroup"
    root.mkdir(exist_ok=True)
    if cpu_max is not None:
        (root / "cpu.max").write_text(cpu_max)
    if v1 is not None:
        (root / "cpu").mkdir(exist_ok=True)
        (root / "cpu" / "cpu.cfs_quota_us").write_text("%d\n" % v1[0])
        (root / "cpu" / "cpu.cfs_period_us").write_text("%d\n" % v1[1])
    e = {k: v for k, v in os.environ.items() if k not in ("LOCAL_WORLD_SIZE", "SNPGPU_LOCAL_RANKS", "SNPGPU_MAX_CPU_CORES")}
    e.update(SNPGPU_CGROUP_ROOT=str(root), PROBE_AFFINITY=str(affinity), **{k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", PROBE], env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.splitlines()[-1])


def test_sixteen_cores_shared_by_eight_ranks(tmp_path):
    """a box that shows many CPUs and grants 16 (the quota), 8 ranks under torch.distributed.run: 2 CPUs per rank, at
    most 2 readers and 1 writer each — 8 + 8 threads in all, not 64 readers plus writer pools."""
    have = len(os.sched_getaffinity(0))
    b = _probe(tmp_path, have, cpu_max="1600000 100000\n", LOCAL_WORLD_SIZE=8)["first"]
    assert b["affinity_cpus"] == have and b["quota_cpus"] == 16 and b["local_ranks"] == 8
    assert b["usable_cpus"] == min(have, 16) and b["budget"] == max(1, min(have, 16) // 8)
    assert 1 <= b["readers"] <= 2 and b["writers"] == 1
    assert 8 * b["readers"] <= 16


def test_affinity_mask_quota_and_cap(tmp_path):
    have = len(os.sched_getaffinity(0))
    small = max(1, have // 2)
    # the affinity mask alone (no quota: "max"), one rank
    b = _probe(tmp_path, small, cpu_max="max 100000\n")["first"]
    assert b["affinity_cpus"] == small and b["quota_cpus"] == 0 and b["usable_cpus"] == small and b["budget"] == small and b["local_ranks"] == 1
    assert b["readers"] == (8 if small >= 32 else small // 2 if small >= 8 else max(1, small - 1))
    # a fractional quota counts as one CPU; cgroup v1 files are read when there is no cpu.max
    assert _probe(tmp_path, have, cpu_max="50000 100000\n")["first"]["usable_cpus"] == 1
    (tmp_path / "cgroup" / "cpu.max").unlink()
    b = _probe(tmp_path, have, v1=(300000, 100000))["first"]
    assert b["quota_cpus"] == 3 and b["usable_cpus"] == min(have, 3)
    assert _probe(tmp_path, have, v1=(-1, 100000))["first"]["quota_cpus"] == 0
    # MaxCpuCores: the environment, then the setter (which wins); ranks: SNPGPU_LOCAL_RANKS beats LOCAL_WORLD_SIZE, the setter beats both
    got = _probe(tmp_path, have, SNPGPU_MAX_CPU_CORES=2, SNPGPU_LOCAL_RANKS=2, LOCAL_WORLD_SIZE=64, PROBE_SET_RANKS=1, PROBE_SET_CORES=1)
    assert got["first"]["max_cpu_cores"] == 2 and got["first"]["usable_cpus"] == min(have, 2) and got["first"]["local_ranks"] == 2
    assert got["first"]["budget"] == max(1, min(have, 2) // 2)
    assert got["after_set_ranks"]["local_ranks"] == 1 and got["after_set_ranks"]["budget"] == min(have, 2)
    assert got["after_set_cores"]["max_cpu_cores"] == 1 and got["after_set_cores"]["budget"] == 1 and got["after_set_cores"]["readers"] == 1
    # nonsense in the environment is no cap
    assert _probe(tmp_path, have, SNPGPU_MAX_CPU_CORES="lots", LOCAL_WORLD_SIZE="-3")["first"]["local_ranks"] == 1

"""The N > 1 path with the real kernels: bench.py's step under torch.distributed.run, all ranks on the one GPU of the test
box (gloo moves the bytes; on an 8-GPU node the same code runs over RCCL).  The 2- and 3-rank runs must reproduce the 1-rank
answer exactly: site union + carrier lists (C1), packed consensus matrix (C2), consensus rows, and the distance rows each rank
ends up owning after the row-band exchange.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_bench(world, dump, extra, launcher=False, rccl_group_of_one=False, comm=None):
    args = ["bench.py", "--gpus", str(world), "--steps", "2", "--warmup", "1", "--scaling", "strong", "--skip-aux", "--e2e-files", "0", "--site-files", "0",
            "--cpu-samples", "0", "--pipeline-files", "0", "--shape-samples", "0", "--no-live-traffic", "--dump", dump, "--detail", dump + ".detail.json"] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(SNPGPU_BENCH_TEST_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    if rccl_group_of_one:                                               # one rank, one GPU, backend nccl, every collective of the step made
        del env["SNPGPU_BENCH_TEST_ONE_GPU"]
        env.update(SNPGPU_DIST_AT_WORLD_1="1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
        if comm:
            env["SNPGPU_COMM"] = comm
    if world == 1 or not launcher:
        cmd = [sys.executable] + args                                   # plain `python bench.py --gpus N`: bench.py starts its own ranks
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    # the LAST line of stdout is the compact object the driver parses (< 4 KB); the full result is in the detail file, and what the
    # compact line says is what the detail file says
    line = r.stdout.splitlines()[-1]
    assert len(line) < 4096
    compact = json.loads(line)
    with open(dump + ".detail.json") as f:
        detail = json.load(f)
    for key in ("value", "ms_per_step", "n_gpus", "scaling", "steps", "warmup"):
        assert compact[key] == pytest.approx(detail[key], rel=1e-5), key
    assert compact["roofline"]["frac"] == pytest.approx(detail["roofline"]["frac"], rel=1e-5) and compact["comm"]["world_size"] == world
    return detail


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_step_equals_single_rank(tmp_path, world):
    n_total = 300                                                       # 3 x 3 distance tiles, uneven sample shards for 3 ranks
    extra = ["--samples", str(n_total), "--genome", "40000", "--sites", "400", "--vcf-records", "60", "--dist-samples", "700",
             "--dist-sites", "3000", "--dist-reps", "1"]
    one = _run_bench(1, str(tmp_path / "one"), extra)
    many = _run_bench(world, str(tmp_path / "many"), extra, launcher=(world == 3))      # 2 ranks: self-launched; 3: the driver's way
    assert many["n_gpus"] == world and many["scaling"] == "strong" and "world size %d" % world in many["config"]["parallelism"]
    assert many["comm"]["world_size"] == world and many["comm"]["backend"] == "gloo"
    assert ("started its own ranks" in many["comm"]["launcher"]) == (world != 3)      # (8 ranks: six distance tiles, two ranks without one)
    assert len(many["phases_ms_per_step"]["per_rank"]) == world and many["comm_route"] == {"exchanges": "torch.distributed (gloo)"}
    assert one["comm"] == {"backend": None, "world_size": 1, "launcher": "none (one process)", "collectives_per_step": "none"}
    assert many["site_union"] == one["site_union"]
    assert many["secondary"]["value"] > 0
    ref = np.load(str(tmp_path / "one.rank0.npz"))
    assert tuple(ref["band_rows"]) == (0, n_total) and ref["band"].shape == (n_total, n_total)
    full = ref["band"]
    assert np.array_equal(full, full.T) and not full.diagonal().any() and full.max() > 0
    covered = 0
    for r in range(world):
        got = np.load(str(tmp_path / ("many.rank%d.npz" % r)))
        # C1: the same snplist (keys, carrier CSR) on every rank
        for k in ("union", "union_off", "carriers"):
            assert np.array_equal(got[k], ref[k]), (r, k)
        # C2: the same packed matrix on every rank
        assert np.array_equal(got["packed"], ref["packed"]), r
        # this rank's consensus rows and distance rows
        g0, g1 = got["first_sample"]
        assert np.array_equal(got["bases"], ref["bases"][g0:g1])
        lo, hi = got["band_rows"]
        assert np.array_equal(got["band"], full[lo:hi]), r
        covered += hi - lo
    assert covered == n_total


def test_strong_scaling_of_the_distance_step_at_configs4_shape(tmp_path):
    """BASELINE configs[4] (10 000 samples x 200 000 sites) with the tiles dealt to two ranks and the row bands exchanged: the
    rows the ranks end up owning add up — by row and by column — to what the one-rank matrix does."""
    extra = ["--samples", "8", "--genome", "20000", "--sites", "200", "--vcf-records", "20", "--dist-samples", "10000",
             "--dist-sites", "200000", "--dist-reps", "1"]
    one = _run_bench(1, str(tmp_path / "one"), extra)
    two = _run_bench(2, str(tmp_path / "two"), extra)
    assert two["scaling"] == "strong" and "10000 samples x 200000 sites" in two["secondary"]["config"]["workload"]
    assert "row-band exchange included" in two["secondary"]["config"]["workload"]
    assert one["secondary"]["band_checksum"][0] > 0
    assert two["secondary"]["band_checksum"] == one["secondary"]["band_checksum"]
    # per-phase device times of the step are reported for every world size (max over ranks next to rank 0's)
    for run in (one, two):
        ph = run["phases_ms_per_step"]
        assert set(ph["rank0"]) == set(ph["max_over_ranks"]) and all(v >= 0 for v in ph["max_over_ranks"].values())


@pytest.mark.parametrize("route", ["abi", "torch"])
def test_the_step_in_an_rccl_group_of_one(tmp_path, route):
    """RCCL needs a GPU per rank and the test box has one: so the N > 1 step runs over gloo above, and RCCL runs here in a group of
    ONE rank that still makes every collective call of the step on device tensors (SNPGPU_DIST_AT_WORLD_1) — through the library's
    own entry points (csrc/comm.hip: the default over RCCL), and through torch.distributed (SNPGPU_COMM=torch: all_gather_into_tensor
    of counts, padded keys and packed rows, all_to_all_single with split lists).  Same answers as the plain one-process run, array
    for array."""
    n_total = 300
    extra = ["--samples", str(n_total), "--genome", "40000", "--sites", "400", "--vcf-records", "60", "--dist-samples", "700",
             "--dist-sites", "3000", "--dist-reps", "1"]
    one = _run_bench(1, str(tmp_path / "one"), extra)
    grp = _run_bench(1, str(tmp_path / "grp"), extra, rccl_group_of_one=True, comm=route)
    assert grp["comm"]["backend"] == "nccl" and grp["comm"]["world_size"] == 1 and grp["n_gpus"] == 1
    if route == "abi":
        assert "libsnpgpu.so" in grp["comm_route"]["exchanges"] and grp["comm_route"]["world_size_rccl_reports"] == 1 and grp["comm_route"]["rccl_version"] >= 20000
    else:
        assert grp["comm_route"] == {"exchanges": "torch.distributed (nccl)"}
    assert "all-to-all" in grp["comm"]["collectives_per_step"]
    assert grp["site_union"] == one["site_union"]
    assert grp["secondary"]["band_checksum"] == one["secondary"]["band_checksum"] and "row-band exchange included" in grp["secondary"]["config"]["workload"]
    ref = np.load(str(tmp_path / "one.rank0.npz"))
    got = np.load(str(tmp_path / "grp.rank0.npz"))
    assert sorted(ref.files) == sorted(got.files)
    for k in ref.files:
        assert np.array_equal(got[k], ref[k]), k


def _n_devices():
    from snp_pipeline_amd import device as dev
    return dev.device_count()


@pytest.mark.skipif(_n_devices() < 2, reason="RCCL between two ranks needs two GPUs (the test box of this build has one)")
@pytest.mark.parametrize("comm", ["abi", "torch"])
def test_two_ranks_over_rccl_when_two_gpus_are_there(tmp_path, comm):
    """the real inter-rank ncclSend / ncclRecv path.  Skipped on a one-GPU box; on a node with two or more GPUs two ranks
    run the step over RCCL — through the library's communicator, and through torch.distributed — and must reproduce the one-rank
    arrays (uneven sample shards: 301 samples over two ranks)."""
    n_total = 301
    extra = ["--samples", str(n_total), "--genome", "40000", "--sites", "400", "--vcf-records", "60", "--dist-samples", "700",
             "--dist-sites", "3000", "--dist-reps", "1"]
    one = _run_bench(1, str(tmp_path / "one"), extra)
    args = ["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--scaling", "strong", "--skip-aux", "--e2e-files", "0", "--site-files", "0", "--cpu-samples", "0",
            "--pipeline-files", "0", "--shape-samples", "0", "--no-live-traffic", "--dump", str(tmp_path / "two"), "--detail", str(tmp_path / "two.detail.json")] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SNPGPU_BENCH_TEST_ONE_GPU")}
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", SNPGPU_COMM=comm)
    r = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads(r.stdout.splitlines()[-1])
    assert line["n_gpus"] == 2 and line["comm"] == {"backend": "nccl", "world_size": 2}
    assert ("libsnpgpu.so" in line["comm_route"]) == (comm == "abi")
    two = json.load(open(str(tmp_path / "two.detail.json")))
    assert two["site_union"] == one["site_union"] and two["secondary"]["band_checksum"] == one["secondary"]["band_checksum"]
    ref = np.load(str(tmp_path / "one.rank0.npz"))
    for r_ in range(2):
        got = np.load(str(tmp_path / ("two.rank%d.npz" % r_)))
        for k in ("union", "union_off", "carriers", "packed"):
            assert np.array_equal(got[k], ref[k]), (r_, k)
        g0, g1 = got["first_sample"]
        lo, hi = got["band_rows"]
        assert np.array_equal(got["bases"], ref["bases"][g0:g1]) and np.array_equal(got["band"], ref["band"][lo:hi])

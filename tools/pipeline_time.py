#!/usr/bin/env python3
"""Wall time of `hot_path_batch` on N synthetic samples (5 Mbp x 30x by default) written to a sample tree first, with its
phases, next to the pinned host-to-device copy rate and (optionally) the separate subcommands on the same tree.  The helpers
are bench.py's (its pipeline_from_files row is this measurement at the bench's own workload).

    python tools/pipeline_time.py [--samples 125] [--genome 5000000] [--sites 50000] [--separate] [--runs 1] [--no-vcf]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_rows as bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=125)
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--sites", type=int, default=50_000)
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--runs", type=int, default=1)
    ap.add_argument("--separate", action="store_true")
    ap.add_argument("--no-vcf", action="store_true")
    ap.add_argument("--verbose", type=int, default=0)
    ap.add_argument("--dir", type=str, default=None)
    ap.add_argument("--recycle", type=int, default=0, help="GiB of device memory allocated, touched and freed before the tree is written")
    ap.add_argument("--sync", action="store_true", help="os.sync() after writing the tree: the run does not compete with the write-back of 54 GB")
    ap.add_argument("--extra", type=str, default="", help="more options for hot_path_batch, e.g. '--siteCalling existing'")
    ap.add_argument("--resident-frac", type=float, default=0.0,
                    help="after the timed runs: one more run with --residentBytes = this fraction of the pileup bytes (the rest is streamed twice); "
                         "its time and whether every output file equals the fully resident run's")
    ap.add_argument("--probe-open", choices=("none", "stat", "serial", "parallel"), default="none",
                    help="before the first run: time stat / open of every pileup (what does the first open after the write cost?)")
    a = ap.parse_args()
    import torch
    from snp_pipeline_amd import device as dev
    from snp_pipeline_amd import hot_path
    d = dev.Device(0)
    d.use_torch_stream()
    G, S = a.genome, a.sites
    if a.recycle:
        t0 = time.perf_counter()
        x = torch.empty(a.recycle << 30, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        del x
        torch.cuda.empty_cache()
        print("recycle: allocated %d GiB in %.3f s, freed in %.3f s" % (a.recycle, t1 - t0, time.perf_counter() - t1), file=sys.stderr)
    ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda")
    d.synth_reference_dev(1, G, ref.data_ptr())
    torch.cuda.synchronize()
    refh = ref.cpu().numpy()
    rng = np.random.default_rng(2)
    pos = np.unique(rng.choice(np.arange(501, G - 499, dtype=np.int64), size=S, replace=False))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    alt_host = np.zeros(G + 1, dtype=np.uint8)
    alt_host[pos] = acgt[(np.searchsorted(acgt, refh[pos]) + 1 + rng.integers(0, 3, size=len(pos))) % 4]
    alt = torch.from_numpy(alt_host).cuda()
    state = {"buf": None, "cap": 0}

    def sample_bytes(i):
        size = d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), 0, 0, mean_depth=a.depth)
        if size + 256 > state["cap"]:
            state["cap"] = size + (size >> 3) + 256
            state["buf"] = torch.empty(state["cap"], dtype=torch.uint8, device="cuda")
        assert d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), state["buf"].data_ptr(), size, mean_depth=a.depth) == size
        torch.cuda.synchronize()
        return state["buf"][:size].cpu().numpy()

    t0 = time.perf_counter()
    tmpdir, ref_path, dirs_file, dirs, total = bench.write_sample_tree(a.dir or tempfile.gettempdir(), refh, G, sample_bytes, a.samples)
    out = {"samples": a.samples, "pileup_bytes": total, "tree_seconds": time.perf_counter() - t0}
    try:
        out["pinned_h2d_gb_per_sec"] = bench.pinned_h2d_gbps(torch)
        if a.sync:
            t0 = time.perf_counter()
            os.sync()
            out["sync_seconds"] = time.perf_counter() - t0
        if a.probe_open != "none":
            import concurrent.futures
            piles = [os.path.join(sd, "reads.all.pileup") for sd in dirs]
            t0 = time.perf_counter()
            for p in piles:
                os.stat(p)
            out["probe_stat_seconds"] = time.perf_counter() - t0
            if a.probe_open == "serial":                         # whose cost is it: the file's first open anywhere, or this process'?
                import subprocess
                third = len(piles) // 3
                code = "import os,sys,time\nt=time.perf_counter()\nfor p in sys.argv[1:]: os.open(p, os.O_RDONLY)\nprint((time.perf_counter()-t)/max(len(sys.argv)-1,1)*1e3)"
                r = subprocess.run([sys.executable, "-c", code] + piles[:third], capture_output=True, text=True)
                out["probe_open_ms_each_in_a_fresh_process"] = float(r.stdout.strip() or "nan")
                t0 = time.perf_counter()
                for p in piles[:third]:
                    os.close(os.open(p, os.O_RDONLY))
                out["probe_open_ms_each_here_after_that_process"] = (time.perf_counter() - t0) / max(third, 1) * 1e3
                per = []
                for p in piles[third:2 * third]:
                    t0 = time.perf_counter()
                    os.close(os.open(p, os.O_RDONLY))
                    per.append((time.perf_counter() - t0) * 1e3)
                out["probe_open_ms_each_here_first"] = [round(x, 2) for x in per[:12]]
                t0 = time.perf_counter()
                for p in piles[third:2 * third]:
                    with open(p, "rb") as f:
                        f.read(1 << 20)
                out["probe_open_and_read_1MiB_again_ms_each"] = (time.perf_counter() - t0) / max(third, 1) * 1e3
                piles = piles[2 * third:]
            if a.probe_open != "stat":
                t0 = time.perf_counter()
                if a.probe_open == "serial":
                    fds = [os.open(p, os.O_RDONLY) for p in piles]
                else:
                    with concurrent.futures.ThreadPoolExecutor(max_workers=16) as ex:
                        fds = list(ex.map(lambda p: os.open(p, os.O_RDONLY), piles))
                out["probe_open_seconds"] = time.perf_counter() - t0
                for fd in fds:
                    os.close(fd)
        runs = []
        for _ in range(a.runs):
            wall = bench.run_cli(bench.hot_path_line(dirs_file, ref_path, (" --noConsensusVcf" if a.no_vcf else "") + (" " + a.extra if a.extra else "")), verbose=a.verbose)
            st = dict(hot_path.hot_path_batch.last_stats)
            st["cli_seconds"] = wall
            runs.append(st)
        out["runs"] = runs
        best = min(runs, key=lambda r: r["seconds"])
        ideal = total / (out["pinned_h2d_gb_per_sec"] * 1e9)
        out.update({"best_seconds": best["seconds"], "ideal_copy_seconds": ideal, "wall_over_copy": best["seconds"] / ideal,
                    "h2d_equals_file_bytes": best["h2d_bytes"] == total,
                    # what the job costs per sample beyond moving its bytes: Python per sample, the files it writes, launches
                    "host_ms_per_sample_outside_the_copy": (best["seconds"] - ideal) / a.samples * 1e3,
                    "ms_per_sample": best["seconds"] / a.samples * 1e3})
        if a.resident_frac > 0 and not a.no_vcf:
            full = bench.output_digests(tmpdir, dirs)
            budget = int(total * a.resident_frac)
            wall = bench.run_cli(bench.hot_path_line(dirs_file, ref_path, " --residentBytes %d" % budget + (" " + a.extra if a.extra else "")), verbose=a.verbose)
            st = dict(hot_path.hot_path_batch.last_stats)
            part = bench.output_digests(tmpdir, dirs)
            out["partly_resident"] = {"resident_bytes_budget": budget, "resident_files": st["resident_files"], "files": st["files"], "seconds": st["seconds"],
                                      "h2d_bytes": st["h2d_bytes"], "h2d_over_file_bytes": st["h2d_bytes"] / total,
                                      "over_fully_resident": st["seconds"] / best["seconds"], "outputs_identical_to_fully_resident": part == full,
                                      "phases": st["phases"]}
        if a.separate and not a.no_vcf:
            mine = bench.output_digests(tmpdir, dirs)
            for sdir in dirs:
                for name in bench.PER_SAMPLE_FILES:
                    os.remove(os.path.join(sdir, name))
            sep = bench.separate_steps(tmpdir, ref_path, dirs_file)
            theirs = bench.output_digests(tmpdir, dirs)
            out.update({"separate_steps_seconds": sep, "separate_total_seconds": sum(sep.values()), "identical_outputs": mine == theirs,
                        "different": [k for k in mine if mine[k] != theirs[k]]})
        print(json.dumps(out))
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Development helper: wall time of the console script on one full-size sample (5 Mbp x 30x, 50 k sites), as run.py starts it:
a fresh process per call; with and without consensus.vcf.  Usage: python tools/cli_time.py [genome_len]"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from snp_pipeline_amd import device as dev
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    S = G // 100
    d = dev.Device(0)
    d.use_torch_stream()
    ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda")
    d.synth_reference_dev(1, G, ref.data_ptr())
    pos = np.sort(np.random.default_rng(2).choice(np.arange(501, G - 499), size=S, replace=False))
    alt_h = np.zeros(G + 1, dtype=np.uint8)
    alt_h[pos] = ord("A")
    alt = torch.from_numpy(alt_h).cuda()
    n = d.synth_pileup_dev(3, 0, G, ref.data_ptr(), alt.data_ptr(), 0, 0)
    t = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    d.synth_pileup_dev(3, 0, G, ref.data_ptr(), alt.data_ptr(), t.data_ptr(), n + 64)
    tmp = tempfile.mkdtemp(prefix="snpcli_")
    try:
        sdir = os.path.join(tmp, "sample1")
        os.makedirs(sdir)
        with open(os.path.join(sdir, "reads.all.pileup"), "wb") as f:
            f.write(t[:n].cpu().numpy().tobytes())
        with open(os.path.join(tmp, "snplist.txt"), "w") as f:
            for p in pos:
                f.write("synth_chr1\t%d\t1\tsample1\n" % p)
        d.close()
        exe = os.path.join(ROOT, "bin", "cfsan_snp_pipeline")
        base = [sys.executable, exe, "call_consensus", "-v", "0", "-f", "-l", os.path.join(tmp, "snplist.txt"), "-o", os.path.join(sdir, "consensus.fasta"),
                "--minConsDpth", "3"]
        for label, extra in (("fasta only", []), ("fasta + consensus.vcf", ["--vcfFileName", "consensus.vcf"]),
                             ("fasta + vcf, profiled", ["--vcfFileName", "consensus.vcf"])):
            cmd = base + extra + [os.path.join(sdir, "reads.all.pileup")]
            if "profiled" in label:
                cmd = [sys.executable, "-m", "cProfile", "-s", "cumtime"] + cmd[1:]
            for rep in range(2):
                t0 = time.perf_counter()
                r = subprocess.run(cmd, capture_output=True, text=True)
                dt = time.perf_counter() - t0
                assert r.returncode == 0, r.stderr[-2000:]
            print("%-26s %.3f s  (%.1f MB pileup, %d sites)" % (label, dt, n / 1e6, S))
            if label == "fasta + consensus.vcf":                 # where the time goes (SNPGPU_TIMING=1)
                r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, SNPGPU_TIMING="1"))
                print("\n".join(ln for ln in r.stderr.split("\n") if ln.startswith("#")))
            if "profiled" in label:
                print("\n".join(r.stdout.split("\n")[:45]))
        # the same call as a thin client of the per-node service (SNPGPU_SERVICE): the first one starts the server
        svc = os.path.join(tmp, "svc")
        env = dict(os.environ, SNPGPU_SERVICE=svc, SNPGPU_SERVICE_SPAWN="1")
        cmd = base + ["--vcfFileName", "consensus.vcf", os.path.join(sdir, "reads.all.pileup")]
        times = []
        for rep in range(6):
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True, env=env)
            times.append(time.perf_counter() - t0)
            assert r.returncode == 0, r.stderr[-2000:]
        print("through the service        first (starts the server) %.3f s, then %s s" % (times[0], " ".join("%.3f" % x for x in times[1:])))
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, SNPGPU_TIMING="1"))
        print("\n".join(ln for ln in r.stderr.split("\n") if ln.startswith("#")))
        subprocess.run([sys.executable, exe, "serve", "--socketDir", svc, "--stop"], capture_output=True, text=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()

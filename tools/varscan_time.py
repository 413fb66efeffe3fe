#!/usr/bin/env python3
"""Development helper: phase-1 site calling on one synthetic full-size sample written to a file — wall time of
varscan.mpileup2snp (file in the page cache -> var.flt.vcf) over a few passes.
Usage: python tools/varscan_time.py [genome_len] [mean_depth] [passes]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from snp_pipeline_amd import device as dev
    from snp_pipeline_amd import varscan
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    depth = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
    passes = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    d = dev.Device(0)
    d.use_torch_stream()
    ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda")
    d.synth_reference_dev(1, G, ref.data_ptr())
    rng = np.random.default_rng(4)
    S = G // 1000
    pos = np.sort(rng.choice(np.arange(501, G - 499), size=S, replace=False))
    alt_h = np.zeros(G + 1, dtype=np.uint8)
    refh = ref.cpu().numpy()
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    alt_h[pos] = acgt[(np.searchsorted(acgt, refh[pos]) + 1 + rng.integers(0, 3, size=S)) % 4]
    alt = torch.from_numpy(alt_h).cuda()
    n = d.synth_pileup_dev(3, 0, G, ref.data_ptr(), alt.data_ptr(), 0, 0, mean_depth=depth, p_same=1.0, p_other=1.0)
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    d.synth_pileup_dev(3, 0, G, ref.data_ptr(), alt.data_ptr(), buf.data_ptr(), n + 64, mean_depth=depth, p_same=1.0, p_other=1.0)
    tmp = tempfile.mkdtemp(prefix="vs_", dir=os.environ.get("SNPGPU_BENCH_TMP", "/tmp"))
    path = os.path.join(tmp, "reads.all.pileup")
    with open(path, "wb") as f:
        f.write(buf[:n].cpu().numpy().tobytes())
    opts = varscan.Options("--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5")
    for k in range(passes):
        t0 = time.time()
        n_lines, n_rows = varscan.mpileup2snp(d, path, os.path.join(tmp, "var.flt.vcf"), opts)
        dt = time.time() - t0
        t1 = time.time()
        recs, _ = d.varscan_file(path, opts.device_params())
        t_dev = time.time() - t1
        print("pass %d: %d lines, %d sites, %.3f s  = %.1f GB/s file -> var.flt.vcf (the library call alone: %.3f s = %.1f GB/s)" % (k, n_lines, n_rows, dt, n / dt / 1e9, t_dev, n / t_dev / 1e9))
    os.unlink(path)


if __name__ == "__main__":
    main()

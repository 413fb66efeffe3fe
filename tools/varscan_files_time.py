#!/usr/bin/env python3
"""Development helper: N synthetic full-size pileup files through snpgpu_varscan_files (the library call alone) and through
varscan.mpileup2snp_files (with the VCF files written).  Usage: python tools/varscan_files_time.py [n_files] [passes] [site spacing]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from snp_pipeline_amd import device as dev
    from snp_pipeline_amd import varscan
    n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    G = 5_000_000
    d = dev.Device(0)
    d.use_torch_stream()
    ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda")
    d.synth_reference_dev(1, G, ref.data_ptr())
    alt = torch.zeros(G + 1, dtype=torch.uint8, device="cuda")
    spacing = int(sys.argv[3]) if len(sys.argv) > 3 else 1000       # one planted site per `spacing` bases (carried by ~10 % of samples)
    alt[spacing::spacing] = ord("A")
    tmp = tempfile.mkdtemp(prefix="vsf_", dir=os.environ.get("SNPGPU_BENCH_TMP", "/tmp"))
    paths, total = [], 0
    for i in range(n_files):
        n = d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), 0, 0)
        buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), buf.data_ptr(), n + 64)
        path = os.path.join(tmp, "s%d.pileup" % i)
        with open(path, "wb") as f:
            f.write(buf[:n].cpu().numpy().tobytes())
        paths.append(path)
        total += n
    opts = varscan.Options("--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5")
    d.varscan_files(paths[:2], opts.device_params())
    for k in range(passes):
        t0 = time.time()
        res = d.varscan_files(paths, opts.device_params())
        t1 = time.time() - t0
        t0 = time.time()
        varscan.mpileup2snp_files(d, paths, [p + ".vcf" for p in paths], opts)
        t2 = time.time() - t0
        print("pass %d: library call %.3f s = %.1f GB/s (%.1f samples/s); with the VCF files written %.3f s = %.1f GB/s; %d records"
              % (k, t1, total / t1 / 1e9, n_files / t1, t2, total / t2 / 1e9, sum(len(r[0]) for r in res)))
    for p in paths:
        os.unlink(p)


if __name__ == "__main__":
    main()

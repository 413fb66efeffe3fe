#!/usr/bin/env python3
"""Phase-1 site calling on ONE synthetic sample that is already in device memory (snpgpu_varscan_dev): what the two kernels of
csrc/varscan.hip and the line index cost without any file or copy.  Run it under rocprofv3 for the per-kernel numbers:
    rocprofv3 --kernel-trace --stats -d gpurun_out/vs -- python tools/varscan_kernel_time.py [genome_len] [mean_depth] [reps]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from snp_pipeline_amd import device as dev
    from snp_pipeline_amd import varscan
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    depth = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    d = dev.Device(0)
    d.use_torch_stream()
    ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda")
    d.synth_reference_dev(1, G, ref.data_ptr())
    rng = np.random.default_rng(2)
    S = 50_000 * G // 5_000_000
    pos = np.unique(rng.choice(np.arange(501, G - 499), size=S, replace=False))
    refh = ref.cpu().numpy()
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    alt_h = np.zeros(G + 1, dtype=np.uint8)
    alt_h[pos] = acgt[(np.searchsorted(acgt, refh[pos]) + 1 + rng.integers(0, 3, size=len(pos))) % 4]
    alt = torch.from_numpy(alt_h).cuda()
    n = d.synth_pileup_dev(3, 0, G, ref.data_ptr(), alt.data_ptr(), 0, 0, mean_depth=depth)
    buf = torch.empty(n + 8192, dtype=torch.uint8, device="cuda")
    d.synth_pileup_dev(3, 0, G, ref.data_ptr(), alt.data_ptr(), buf.data_ptr(), n, mean_depth=depth)
    torch.cuda.synchronize()
    prm = varscan.Options("--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5").device_params()
    recs, n_lines = d.varscan_dev(buf.data_ptr(), n, prm)
    t = time.perf_counter()
    for _ in range(reps):
        recs, n_lines = d.varscan_dev(buf.data_ptr(), n, prm)
    dt = (time.perf_counter() - t) / reps
    print("%d bytes, %d lines, %d records: %.3f ms per call (two host round trips included) = %.2f TB/s" % (n, n_lines, len(recs), dt * 1e3, n / dt / 1e12))


if __name__ == "__main__":
    main()

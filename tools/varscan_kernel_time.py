#!/usr/bin/env python3
"""Phase-1 site calling on synthetic samples that already are in device memory: ONE sample per call (snpgpu_varscan_dev), then —
with a fourth argument — that many distinct samples in one launch (snpgpu_varscan_batch_dev): what the kernels of csrc/varscan.hip
cost without any file or copy.  HIP events around the launches on their stream; run it under rocprofv3 for the per-kernel numbers:
    rocprofv3 --kernel-trace --stats -d gpurun_out/vs -- python tools/varscan_kernel_time.py [genome_len] [mean_depth] [reps] [batch]
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
    n_batch = int(sys.argv[4]) if len(sys.argv) > 4 else 0
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
    d.kernel_timing(True)
    d.kernel_time_ms(3)
    t = time.perf_counter()
    for _ in range(reps):
        recs, n_lines = d.varscan_dev(buf.data_ptr(), n, prm)
    dt = (time.perf_counter() - t) / reps
    k_ms, k_n = d.kernel_time_ms(3)
    print("%d bytes, %d lines, %d records: %.3f ms per call (host round trips included) = %.2f TB/s; kernels %.1f us = %.2f TB/s (%.3f of 8 TB/s)"
          % (n, n_lines, len(recs), dt * 1e3, n / dt / 1e12, k_ms / max(k_n, 1) * 1e3, n / (k_ms / max(k_n, 1) * 1e-3) / 1e12, n / (k_ms / max(k_n, 1) * 1e-3) / 8e12))
    if n_batch:
        bufs, sizes = [], []
        for i in range(n_batch):
            ni = d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), 0, 0, mean_depth=depth)
            b = torch.empty(ni + 8192, dtype=torch.uint8, device="cuda")
            d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), b.data_ptr(), ni, mean_depth=depth)
            bufs.append(b)
            sizes.append(ni)
        torch.cuda.synchronize()
        ptrs = [b.data_ptr() for b in bufs]
        res = d.varscan_batch_dev(ptrs, sizes, prm)
        assert res[0][0].tobytes() == recs.tobytes() and res[0][1] == n_lines           # (sample 0 again)
        d.kernel_time_ms(3)
        t = time.perf_counter()
        calls = max(reps // 2, 2)
        for _ in range(calls):
            d.varscan_batch_dev(ptrs, sizes, prm)
        dt = (time.perf_counter() - t) / calls
        k_ms, k_n = d.kernel_time_ms(3)                       # (a call makes one launch per dozen 30x samples: k_n launches in all)
        tot = sum(sizes)
        per_call = k_ms / calls
        print("batch of %d samples, %d bytes: %.3f ms per call; kernels %.1f us per call (%d launches) = %.1f us per sample = %.2f TB/s (%.3f of 8 TB/s)"
              % (n_batch, tot, dt * 1e3, per_call * 1e3, k_n // calls, per_call * 1e3 / n_batch, tot / (per_call * 1e-3) / 1e12, tot / (per_call * 1e-3) / 8e12))
    d.kernel_timing(False)


if __name__ == "__main__":
    main()

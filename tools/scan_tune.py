#!/usr/bin/env python3
"""Development helper: time k_scan_wave / k_call_sites on a few device-generated samples.
Usage: python tools/scan_tune.py [n_samples] [genome_len] [batch|single] [mean_depth] [contig name]   (knobs via SNPGPU_SCAN_* env vars)
"batch": all samples through one call of the batch entry point (one scan launch, one call launch)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from snp_pipeline_amd import _lib as L
    if os.environ.get("SNPGPU_TUNE_LIB"):                       # A/B timing against another build (tools/ab_build.sh)
        L.LIB_PATH = os.path.abspath(os.environ["SNPGPU_TUNE_LIB"])
    from snp_pipeline_amd import device as dev
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
    depth = float(sys.argv[4]) if len(sys.argv) > 4 else 30.0
    contig = sys.argv[5].encode() if len(sys.argv) > 5 else b"synth_chr1"
    S = G // 100
    d = dev.Device(0)
    d.use_torch_stream()
    ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda")
    d.synth_reference_dev(1, G, ref.data_ptr())
    pos = np.sort(np.random.default_rng(2).choice(np.arange(501, G - 499), size=S, replace=False))
    alt_h = np.zeros(G + 1, dtype=np.uint8)
    alt_h[pos] = ord("A")
    alt = torch.from_numpy(alt_h).cuda()
    bufs, sizes = [], []
    for i in range(B):
        n = d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), 0, 0, mean_depth=depth, contig=contig)
        t = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), t.data_ptr(), n + 64, mean_depth=depth, contig=contig)
        bufs.append(t)
        sizes.append(n)
    ss = d.siteset([(contig, int(p)) for p in pos], [1] * S)
    prm = dev.make_params(0, 0.6, 3, 0, 0.0)
    bases = torch.empty((B, S), dtype=torch.uint8, device="cuda")
    filt = torch.empty((B, S), dtype=torch.uint8, device="cuda")
    status = torch.empty((B, 4), dtype=torch.int64, device="cuda")

    batch = len(sys.argv) > 3 and sys.argv[3] == "batch"
    ptrs = [t.data_ptr() for t in bufs]
    base = min(ptrs)
    offs = np.array([p - base for p in ptrs], dtype=np.uint64)

    counts = torch.empty(S * 128, dtype=torch.uint8, device="cuda") if os.environ.get("SNPGPU_TUNE_COUNTS") == "1" else None

    def run():
        if counts is not None:                                  # the consensus.vcf path: per-site records
            for i in range(B):
                d.call_consensus_dev(ss, bufs[i].data_ptr(), sizes[i], prm, bases[i].data_ptr(), filt[i].data_ptr(), status[i].data_ptr(),
                                     d_counts=counts.data_ptr())
            return
        if batch:
            d.call_consensus_batch_dev(ss, base, offs, prm, bases.data_ptr(), filt.data_ptr(), status.data_ptr(), sizes=sizes)
            return
        for i in range(B):
            d.call_consensus_dev(ss, bufs[i].data_ptr(), sizes[i], prm, bases[i].data_ptr(), filt[i].data_ptr(), status[i].data_ptr())
    run()
    torch.cuda.synchronize()
    d.kernel_timing(True)
    d.kernel_time_ms(0), d.kernel_time_ms(1)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    sm, sn = d.kernel_time_ms(0)
    cm, cn = d.kernel_time_ms(1)
    st = status.cpu().numpy()
    gbs = sum(sizes) * 3 / (sm * 1e-3) / 1e9
    if batch:
        sn, cn = sn * B, cn * B
    print("variant waves=%s blocks=%s : scan %.3f ms/sample  %.0f GB/s (%.1f%% of 8 TB/s) | call %.3f ms/sample | lines %d matched %d err %s | checksum %d"
          % (os.environ.get("SNPGPU_SCAN_WAVES", "-"), os.environ.get("SNPGPU_SCAN_BLOCKS_PER_CU", "-"), sm / sn, gbs, gbs / 80,
             cm / cn, st[0, 1], st[0, 2], st[0, 0] != -1, int(bases.to(torch.int64).sum().item())))


if __name__ == "__main__":
    main()

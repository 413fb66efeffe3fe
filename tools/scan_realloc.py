#!/usr/bin/env python3
"""Development helper: does the scan rate of the headline launch depend on WHERE the samples lie in device memory?  One process;
the same 125 device-generated samples are freed and generated again several times, with the allocator's cache emptied and a pad of a
different size kept alive in between, and the launch is timed each time.  Usage: python tools/scan_realloc.py [n] [depth] [rounds]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from snp_pipeline_amd import _lib as L
    if os.environ.get("SNPGPU_TUNE_LIB"):
        L.LIB_PATH = os.path.abspath(os.environ["SNPGPU_TUNE_LIB"])
    from snp_pipeline_amd import device as dev
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 125
    depth = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    one_arena = os.environ.get("REALLOC_ARENA") == "1"
    G = int(os.environ.get("SWEEP_GENOME", "5000000"))              # (toy sizes for the test of this helper)
    S = G // 100
    contig = b"synth_chr1"
    d = dev.Device(0)
    d.use_torch_stream()
    ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda")
    d.synth_reference_dev(1, G, ref.data_ptr())
    pos = np.sort(np.random.default_rng(2).choice(np.arange(501, G - 499), size=S, replace=False))
    alt_h = np.zeros(G + 1, dtype=np.uint8)
    alt_h[pos] = ord("A")
    alt = torch.from_numpy(alt_h).cuda()
    ss = d.siteset([(contig, int(p)) for p in pos], [1] * S)
    prm = dev.make_params(0, 0.6, 3, 0, 0.0)
    bases = torch.empty((B, S), dtype=torch.uint8, device="cuda")
    filt = torch.empty((B, S), dtype=torch.uint8, device="cuda")
    status = torch.empty((B, 4), dtype=torch.int64, device="cuda")
    sizes = [d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), 0, 0, mean_depth=depth, contig=contig) for i in range(B)]
    pads = []
    for r in range(rounds):
        if one_arena:                                           # one allocation for the whole batch, samples 4 KiB aligned inside it
            offs_in = np.cumsum([0] + [(n + 64 + 4095) // 4096 * 4096 for n in sizes])
            arena = torch.empty(int(offs_in[-1]), dtype=torch.uint8, device="cuda")
            bufs = [arena[int(offs_in[i]):int(offs_in[i]) + sizes[i] + 64] for i in range(B)]
        else:
            bufs = [torch.empty(n + 64, dtype=torch.uint8, device="cuda") for n in sizes]
        for i in range(B):
            d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), bufs[i].data_ptr(), sizes[i] + 64, mean_depth=depth, contig=contig)
        ptrs = [t.data_ptr() for t in bufs]
        base = min(ptrs)
        offs = np.array([p - base for p in ptrs], dtype=np.uint64)
        run = lambda: d.call_consensus_batch_dev(ss, base, offs, prm, bases.data_ptr(), filt.data_ptr(), status.data_ptr(), sizes=sizes)
        run()
        torch.cuda.synchronize()
        d.kernel_timing(True)
        d.kernel_time_ms(0), d.kernel_time_ms(1)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        sm, sn = d.kernel_time_ms(0)
        gbs = sum(sizes) * 3 / (sm * 1e-3) / 1e9
        print("round %d: %6.0f GB/s  %.1f %%   first sample at 0x%x, span %.1f GB, pads %d MB" % (r, gbs, gbs / 80, base, (max(ptrs) - base) / 1e9,
                                                                                        sum(p.numel() for p in pads) >> 20), flush=True)
        del bufs, run
        if one_arena:
            del arena
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        pads.append(torch.empty((r + 1) * 37 * (1 << 20) + 4096 * (r + 1), dtype=torch.uint8, device="cuda"))


if __name__ == "__main__":
    main()

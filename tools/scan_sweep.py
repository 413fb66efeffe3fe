#!/usr/bin/env python3
"""Development helper (tuning build, SNPGPU_TUNE_LIB): time k_scan_wave under many SNPGPU_SCAN_* settings in ONE process, on the same
device-generated samples.  Usage: python tools/scan_sweep.py n_samples depth "WAVES=12 SHARE=300,270,235,195" "WAVES=16" ...
Every setting is timed `reps` times (SWEEP_REPS, default 2), the settings interleaved."""
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
    B, depth = int(sys.argv[1]), float(sys.argv[2])
    settings = sys.argv[3:] or [""]
    G = int(os.environ.get("SWEEP_GENOME", "5000000"))              # (toy sizes for the test of this helper)
    S = G // 100
    contig = b"synth_chr1"
    os.environ["SNPGPU_SCAN_RELOAD"] = "1"
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
    ptrs = [t.data_ptr() for t in bufs]
    base = min(ptrs)
    offs = np.array([p - base for p in ptrs], dtype=np.uint64)

    def run():
        d.call_consensus_batch_dev(ss, base, offs, prm, bases.data_ptr(), filt.data_ptr(), status.data_ptr(), sizes=sizes)

    first = None
    for rep in range(int(os.environ.get("SWEEP_REPS", "2"))):
        for st in settings:
            for k in [k for k in os.environ if k.startswith("SNPGPU_SCAN_") and k != "SNPGPU_SCAN_RELOAD"]:
                del os.environ[k]
            for kv in st.split():
                k, v = kv.split("=")
                os.environ["SNPGPU_SCAN_" + k] = v
            run()
            torch.cuda.synchronize()
            d.kernel_timing(True)
            d.kernel_time_ms(0), d.kernel_time_ms(1)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            sm, sn = d.kernel_time_ms(0)
            gbs = sum(sizes) * 3 / (sm * 1e-3) / 1e9
            chk = (int(bases.to(torch.int64).sum().item()), int(status[:, 1].sum().item()), int(status[:, 2].sum().item()))
            first = first or chk
            print("%-44s %6.0f GB/s  %.1f %%  %s" % (st or "(default)", gbs, gbs / 80, "" if chk == first else "RESULTS DIFFER %s" % (chk,)), flush=True)


if __name__ == "__main__":
    main()

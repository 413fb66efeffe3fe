#!/usr/bin/env python3
"""Development helper: time k_distance alone.  Usage: python tools/dist_tune.py [n_samples] [n_sites] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from snp_pipeline_amd import _lib as L
    if os.environ.get("SNPGPU_TUNE_LIB"):
        L.LIB_PATH = os.path.abspath(os.environ["SNPGPU_TUNE_LIB"])
    from snp_pipeline_amd import device as dev
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    d = dev.Device(0)
    d.use_torch_stream()
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    lut = torch.tensor(list(b"ACGT-"), dtype=torch.uint8, device="cuda")
    sym = torch.empty((n, s), dtype=torch.uint8, device="cuda")
    chunk = max(1, (1 << 27) // s)
    for r0 in range(0, n, chunk):
        r1 = min(n, r0 + chunk)
        sym[r0:r1] = lut[torch.randint(0, 5, ((r1 - r0) * s,), device="cuda", generator=g)].view(r1 - r0, s)
    pk = torch.empty((n, d.packed_row_bytes(s)), dtype=torch.uint8, device="cuda")
    d.pack_matrix_dev(sym.data_ptr(), n, s, s, pk.data_ptr())
    dm = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    d.distance_packed_dev(pk.data_ptr(), n, s, dm.data_ptr(), 0, 1)
    torch.cuda.synchronize()
    d.kernel_timing(True)
    d.kernel_time_ms(2)
    for _ in range(reps):
        d.distance_packed_dev(pk.data_ptr(), n, s, dm.data_ptr(), 0, 1)
    torch.cuda.synchronize()
    ms, k = d.kernel_time_ms(2)
    ms /= max(k, 1)
    pairs = n * (n - 1) / 2
    peak = 256 * 64 * 2.4e9                                     # integer VALU lane-ops/s: 4 SIMDs x 16 lanes per CU
    # spot check against torch on a few rows
    rows = [0, 1, n // 2, n - 1]
    a = sym[rows].to(torch.int16)
    ok = True
    for i, r in enumerate(rows):
        va = (a[i] != ord("-"))
        for c in (2, n // 3, n - 2):
            b = sym[c].to(torch.int16)
            want = int(((a[i] != b) & va & (b != ord("-"))).sum().item())
            ok = ok and want == int(dm[r, c].item()) == int(dm[c, r].item())
    print("k_distance %d x %d: %.2f ms  %.3g pairs/s  %.3g site-compares/s  VALU %.1f%% of peak (4 ops / 32 compares)  check %s"
          % (n, s, ms, pairs / (ms * 1e-3), pairs * s / (ms * 1e-3), 100 * (pairs * s / 32 * 4 / (ms * 1e-3)) / peak, ok))


if __name__ == "__main__":
    main()

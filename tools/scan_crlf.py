#!/usr/bin/env python3
"""Development helper: scan rate on a CR LF copy of a device-generated sample next to the LF original."""
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
    G = 5_000_000
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
    lf = t[:n].cpu().numpy().tobytes()
    crlf = lf.replace(b"\n", b"\r\n")
    ss = d.siteset([(b"synth_chr1", int(p)) for p in pos], [1] * S)
    prm = dev.make_params(0, 0.6, 3, 0, 0.0)
    B = 8
    res = {}
    for name, data in (("LF", lf), ("CRLF", crlf)):
        one = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        step = (len(data) + 255) // 256 * 256
        buf = torch.empty(B * step + 64, dtype=torch.uint8, device="cuda")
        for i in range(B):
            buf[i * step:i * step + len(data)] = one
        offs = np.arange(B, dtype=np.uint64) * step
        sizes = np.full(B, len(data), dtype=np.uint64)
        bases = torch.empty((B, S), dtype=torch.uint8, device="cuda")
        filt = torch.empty((B, S), dtype=torch.uint8, device="cuda")
        status = torch.empty((B, 4), dtype=torch.int64, device="cuda")
        run = lambda: d.call_consensus_batch_dev(ss, buf.data_ptr(), offs, prm, bases.data_ptr(), filt.data_ptr(), status.data_ptr(), sizes=sizes)
        run()
        torch.cuda.synchronize()
        d.kernel_timing(True)
        d.kernel_time_ms(0)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        sm, sn = d.kernel_time_ms(0)
        d.kernel_timing(False)
        res[name] = bytes(bases[0].cpu().numpy())
        print("%-4s %.1f MB/sample: scan %.0f GB/s, lines %d, status0 %d" % (name, len(data) / 1e6, B * len(data) * 3 / (sm * 1e-3) / 1e9, int(status[0, 1]), int(status[0, 0])))
    print("same consensus:", res["LF"] == res["CRLF"])


if __name__ == "__main__":
    main()

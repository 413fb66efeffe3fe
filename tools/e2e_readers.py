#!/usr/bin/env python3
"""Development helper: end-to-end streamed rate against the number of reader threads, files written by this process
(page cache wherever the kernel put it).  Usage: python tools/e2e_readers.py [n_files]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from snp_pipeline_amd import _lib as L
    from snp_pipeline_amd import device as dev
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
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
    tmpdir = tempfile.mkdtemp(prefix="snpe2e_")
    paths, total = [], 0
    try:
        for i in range(B):
            n = d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), 0, 0)
            buf = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
            d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), buf.data_ptr(), n + 16)
            path = os.path.join(tmpdir, "s%d.pileup" % i)
            with open(path, "wb") as f:
                f.write(buf[:n].cpu().numpy().tobytes())
            paths.append(path)
            total += n
        ss = d.siteset([(b"synth_chr1", int(p)) for p in pos], [L.SITE_IN_SNPLIST] * S)
        prm = dev.make_params(0, 0.6, 3, 0, 0.0)
        d._check(d.lib.snpgpu_ctx_reset_stream(d.ctx))
        d.call_consensus_files(ss, paths[:1], prm)
        for readers, staging in ((4, 8), (8, 12), (12, 16), (16, 20), (24, 28), (32, 36), (8, 12), (16, 20)):
            rates = []
            for _ in range(3):
                _, _, st = d.call_consensus_files(ss, paths, prm, n_readers=readers, n_staging=staging)
                rates.append(st.bytes / st.seconds / 1e9)
            print("readers %2d staging %2d: %s GB/s" % (readers, staging, " ".join("%.1f" % r for r in rates)), flush=True)
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)


if __name__ == "__main__":
    main()

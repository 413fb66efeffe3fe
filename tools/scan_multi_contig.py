#!/usr/bin/env python3
"""Development helper: scan rate on samples made of many contigs (device-generated pieces with different contig names
laid back to back), some of them without any site.  Usage: python tools/scan_multi_contig.py [n_contigs] [contig_len] [n_samples] [names] [all]
names: mixed (default: every other contig "NODE_<k>_len_<G>", the others "ctg<kkk>"), short (all "ctg<kkk>"), long (all "NODE_<k>_len_<G>"),
draft (SPAdes style: "NODE_<k>_length_<G>_cov_<x.y>", 27-30 bytes) — which of name length and contig changes costs what."""
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
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 125_000
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    d = dev.Device(0)
    d.use_torch_stream()
    ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda")
    d.synth_reference_dev(1, G, ref.data_ptr())
    pos = np.sort(np.random.default_rng(2).choice(np.arange(51, G - 49), size=G // 100, replace=False))
    alt_h = np.zeros(G + 1, dtype=np.uint8)
    alt_h[pos] = ord("A")
    alt = torch.from_numpy(alt_h).cuda()
    style = sys.argv[4] if len(sys.argv) > 4 else "mixed"
    long_name = lambda c: ("NODE_%d_len_%d" % (c + 1, G)).encode()      # noqa: E731
    short_name = lambda c: ("ctg%03d" % c).encode()                     # noqa: E731
    names = [{"mixed": long_name(c) if c % 2 else short_name(c), "short": short_name(c), "long": long_name(c),
              "draft": ("NODE_%d_length_%d_cov_%.1f" % (c + 1, G, 10 + (c * 37 % 400) / 10.0)).encode()}[style] for c in range(C)]
    every = len(sys.argv) > 5 and sys.argv[5] == "all"                               # 5th argument "all": every contig has sites
    keys = [(names[c], int(p)) for c in range(C) if every or c % 5 != 3 for p in pos]          # (default) every fifth contig has no site
    sizes = [[d.synth_pileup_dev(3, s * C + c, G, ref.data_ptr(), alt.data_ptr(), 0, 0, contig=names[c]) for c in range(C)] for s in range(B)]
    total = sum(sum(x) for x in sizes)
    buf = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
    offs, lens, o = [], [], 0
    for s in range(B):
        offs.append(o)
        for c in range(C):
            n = d.synth_pileup_dev(3, s * C + c, G, ref.data_ptr(), alt.data_ptr(), buf.data_ptr() + o, sizes[s][c], contig=names[c])
            o += n
        lens.append(o - offs[-1])
    ss = d.siteset(keys, [1] * len(keys))
    prm = dev.make_params(0, 0.6, 3, 0, 0.0)
    S = len(keys)
    bases = torch.empty((B, S), dtype=torch.uint8, device="cuda")
    filt = torch.empty((B, S), dtype=torch.uint8, device="cuda")
    status = torch.empty((B, 4), dtype=torch.int64, device="cuda")

    def run():
        d.call_consensus_batch_dev(ss, buf.data_ptr(), np.asarray(offs, dtype=np.uint64), prm, bases.data_ptr(), filt.data_ptr(),
                                   status.data_ptr(), sizes=np.asarray(lens, dtype=np.uint64))
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
    print("names %s (%d-%d bytes): " % (style, min(len(x) for x in names), max(len(x) for x in names)), end="")
    print("%d samples x %d contigs x %d bp (%.2f GB): scan %.3f ms  %.0f GB/s | call %.3f ms | lines %d matched %d of %d sites, err %s"
          % (B, C, G, total / 1e9, sm / sn, total / (sm / sn * 1e-3) / 1e9, cm / cn, st[0, 1], st[0, 2], S, (st[:, 0] != -1).any()))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Development helper: the scan's rate by the number of samples in the launch and by WHICH samples of a resident set are scanned
(the first B, every other one, the last 64, the first 64 again after other launches), six launches each, in one process.
profiles/r5/scan_batch_sizes_*.log.   Usage: python tools/scan_batch_sizes.py [depth] [samples resident]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snp_pipeline_amd import device as dev, _lib as L
G = int(os.environ.get("SWEEP_GENOME", "5000000"))              # (toy sizes for the test of this helper)
depth = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 125
d = dev.Device(0); d.use_torch_stream()
ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda"); d.synth_reference_dev(1, G, ref.data_ptr())
pos = np.sort(np.random.default_rng(2).choice(np.arange(501, G - 499), size=G // 100, replace=False))
alt_h = np.zeros(G + 1, dtype=np.uint8); alt_h[pos] = ord("A"); alt = torch.from_numpy(alt_h).cuda()
S = len(pos)
ss = d.siteset([(b"synth_chr1", int(p)) for p in pos], [L.SITE_IN_SNPLIST] * S)
prm = dev.make_params(0, 0.6, 3, 0, 0.0)
sizes = [d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), 0, 0, mean_depth=depth) for i in range(NB)]
offs = np.concatenate(([0], np.cumsum([(n + 255) // 256 * 256 for n in sizes])))
buf = torch.empty(int(offs[-1]) + 8192, dtype=torch.uint8, device="cuda")
for i in range(NB):
    d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), buf.data_ptr() + int(offs[i]), sizes[i], mean_depth=depth)
torch.cuda.synchronize()
bases = torch.empty((NB, S), dtype=torch.uint8, device="cuda"); filt = torch.empty((NB, S), dtype=torch.uint8, device="cuda")
status = torch.empty((NB, 4), dtype=torch.int64, device="cuda")
def measure(idx, label, reps=6):
    o = np.asarray([offs[i] for i in idx], dtype=np.uint64); sz = np.asarray([sizes[i] for i in idx], dtype=np.uint64)
    run = lambda: d.call_consensus_batch_dev(ss, buf.data_ptr(), o, prm, bases.data_ptr(), filt.data_ptr(), status.data_ptr(), sizes=sz)
    run(); torch.cuda.synchronize()
    d.kernel_timing(True); d.kernel_time_ms(0)
    each = []
    for _ in range(reps):
        run(); torch.cuda.synchronize()
        ms, n = d.kernel_time_ms(0); each.append(ms / max(n, 1))
    d.kernel_timing(False)
    nb = int(sz.sum())
    print("%-28s B %3d  %6.2f GB  per launch ms %s  best %.4f of peak  mean %.4f" % (label, len(idx), nb / 1e9, " ".join("%.3f" % x for x in each),
          nb / (min(each) * 1e-3) / 8e12, nb / (np.mean(each) * 1e-3) / 8e12), flush=True)
for B in sorted(set(min(b, NB) for b in (16, 32, 48, 64, 96, NB))):
    measure(list(range(B)), "first %d" % B)
measure(list(range(0, NB, 2)), "every other")
measure(list(range(max(0, NB - 64), NB)), "last 64")
measure(list(range(min(64, NB))), "first 64 again")
measure(list(range(NB)), "all again", reps=12)

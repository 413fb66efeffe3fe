#!/usr/bin/env python3
"""Development helper: is the scan's rate a matter of WHERE things lie in device memory — and of which things?  One process, one
launch shape (NB samples at one depth), timed (1) twice as it is, (2) with six fresh library contexts (scratch, site set, tables
allocated anew; the input untouched), (3) with the input freed and generated again six times (a pad of another size kept alive in
between, so the allocator hands out other pages behind what is often the same virtual address).  Round 5 found the "placement modes"
of rounds 3-4 with it: profiles/r5/scan_placement_*.log.  (A context's FIRST measurement follows host work — the GPU's clocks are
still coming up, which shows at 8x / 15x where the parse is the limit.)   Usage: python tools/scan_placement.py [depth] [samples]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snp_pipeline_amd import device as dev, _lib as L
G = int(os.environ.get("SWEEP_GENOME", "5000000"))              # (toy sizes for the test of this helper)
depth = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 64
S = G // 100
pos = np.sort(np.random.default_rng(2).choice(np.arange(501, G - 499), size=S, replace=False))
keys = [(b"synth_chr1", int(p)) for p in pos]
prm = dev.make_params(0, 0.6, 3, 0, 0.0)
d = dev.Device(0); d.use_torch_stream()
ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda"); d.synth_reference_dev(1, G, ref.data_ptr())
alt_h = np.zeros(G + 1, dtype=np.uint8); alt_h[pos] = ord("A"); alt = torch.from_numpy(alt_h).cuda()
sizes = [d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), 0, 0, mean_depth=depth) for i in range(NB)]
offs = np.concatenate(([0], np.cumsum([(n + 255) // 256 * 256 for n in sizes])))
def gen(dv):
    buf = torch.empty(int(offs[-1]) + 8192, dtype=torch.uint8, device="cuda")
    for i in range(NB):
        dv.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), buf.data_ptr() + int(offs[i]), sizes[i], mean_depth=depth)
    torch.cuda.synchronize()
    return buf
bases = torch.empty((NB, S), dtype=torch.uint8, device="cuda"); filt = torch.empty((NB, S), dtype=torch.uint8, device="cuda")
status = torch.empty((NB, 4), dtype=torch.int64, device="cuda")
o = np.asarray(offs[:-1], dtype=np.uint64); sz = np.asarray(sizes, dtype=np.uint64)
def measure(dv, ss, buf, label):
    if os.environ.get("NO_CHECK"): pass
    run = lambda: dv.call_consensus_batch_dev(ss, buf.data_ptr(), o, prm, bases.data_ptr(), filt.data_ptr(), status.data_ptr(), sizes=sz)
    run(); torch.cuda.synchronize()
    dv.kernel_timing(True); dv.kernel_time_ms(0)
    for _ in range(4): run()
    torch.cuda.synchronize()
    ms, n = dv.kernel_time_ms(0); dv.kernel_timing(False)
    print("%-44s %.3f ms  %.4f of peak   input at 0x%x" % (label, ms / n, int(sz.sum()) / (ms / n * 1e-3) / 8e12, buf.data_ptr()), flush=True)
buf = gen(d)
ss = d.siteset(keys, [L.SITE_IN_SNPLIST] * S)
measure(d, ss, buf, "ctx 0, input 0")
measure(d, ss, buf, "ctx 0, input 0 (again)")
pads = []
print("-- new library contexts (scratch and site set allocated anew), same input")
for t in range(6):
    pads.append(torch.empty((t + 1) * 53 * (1 << 20) + 4096 * t, dtype=torch.uint8, device="cuda"))
    d2 = dev.Device(0); d2.use_torch_stream()
    ss2 = d2.siteset(keys, [L.SITE_IN_SNPLIST] * S)
    measure(d2, ss2, buf, "ctx %d, input 0" % (t + 1))
    ss2.close() if hasattr(ss2, "close") else None
    d2.close() if hasattr(d2, "close") else None
print("-- the first context again, inputs generated anew")
measure(d, ss, buf, "ctx 0, input 0 (again)")
for t in range(6):
    del buf
    torch.cuda.empty_cache()
    pads.append(torch.empty((t + 1) * 71 * (1 << 20) + 4096 * t, dtype=torch.uint8, device="cuda"))
    buf = gen(d)
    measure(d, ss, buf, "ctx 0, input %d" % (t + 1))

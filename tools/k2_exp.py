#!/usr/bin/env python3
"""Scan and call-kernel time of one bench step (125 samples x 5 Mbp x 30x, 50 k sites) with the library named by SNPGPU_LIB: the A/B
harness of the K2 experiment builds (tools/variant_build.sh <name> -DCALL_EXP=... / -DCALL_EXP_ALIGN128; their results are wrong, their
timing is what is asked).  Usage on the GPU box: SNPGPU_LIB=tools/ab/libsnpgpu_<name>.so python tools/k2_exp.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import numpy as np, torch
from snp_pipeline_amd import device as dev, _lib as L
G = int(os.environ.get("SWEEP_GENOME", "5000000")); NB = int(os.environ.get("K2_SAMPLES", "125")); depth = 30.0
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
o = np.asarray(offs[:-1], dtype=np.uint64); sz = np.asarray(sizes, dtype=np.uint64)
run = lambda: d.call_consensus_batch_dev(ss, buf.data_ptr(), o, prm, bases.data_ptr(), filt.data_ptr(), status.data_ptr(), sizes=sz)
run(); torch.cuda.synchronize()
d.kernel_timing(True); d.kernel_time_ms(0); d.kernel_time_ms(1)
for _ in range(5): run()
torch.cuda.synchronize()
s_ms, s_n = d.kernel_time_ms(0); c_ms, c_n = d.kernel_time_ms(1)
print("%s: scan %.3f ms, call kernels %.3f ms per step" % (os.environ.get("SNPGPU_LIB", "product"), s_ms / s_n, c_ms / 5))

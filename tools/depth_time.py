import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snp_pipeline_amd import device as dev
G, S = 5_000_000, 50_000
d = dev.Device(0); d.use_torch_stream()
ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda"); d.synth_reference_dev(1, G, ref.data_ptr())
pos = np.sort(np.random.default_rng(2).choice(np.arange(501, G - 499), size=S, replace=False))
alt_h = np.zeros(G + 1, dtype=np.uint8); alt_h[pos] = ord("A"); alt = torch.from_numpy(alt_h).cuda()
n = d.synth_pileup_dev(3, 0, G, ref.data_ptr(), alt.data_ptr(), 0, 0)
t = torch.empty(n + 64, dtype=torch.uint8, device="cuda"); d.synth_pileup_dev(3, 0, G, ref.data_ptr(), alt.data_ptr(), t.data_ptr(), n + 64)
ss = d.siteset([(b"synth_chr1", int(p)) for p in pos], [1] * S)
prm = dev.make_params(0, 0.6, 3, 0, 0.0)
b = torch.empty(S, dtype=torch.uint8, device="cuda"); f = torch.empty(S, dtype=torch.uint8, device="cuda"); st = torch.zeros(4, dtype=torch.int64, device="cuda")
for depth in (False, True):
    d.call_consensus_dev(ss, t.data_ptr(), n, prm, b.data_ptr(), f.data_ptr(), st.data_ptr(), want_depth_sum=depth)
    torch.cuda.synchronize(); d.kernel_timing(True); d.kernel_time_ms(0)
    for _ in range(5): d.call_consensus_dev(ss, t.data_ptr(), n, prm, b.data_ptr(), f.data_ptr(), st.data_ptr(), want_depth_sum=depth)
    torch.cuda.synchronize(); ms, k = d.kernel_time_ms(0); d.kernel_timing(False)
    print("want_depth_sum=%s: scan %.3f ms  %.0f GB/s  depth_sum %d lines %d" % (depth, ms / k, n / (ms / k * 1e-3) / 1e9, int(st[3]), int(st[1])))
# reference value of the depth sum: count via torch? (mean depth 30 x 5 M lines)

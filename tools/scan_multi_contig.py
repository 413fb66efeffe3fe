import sys, time; sys.path.insert(0, "/root/repo")
import torch
from oracle import fuzz
from snp_pipeline_amd import device as dev
d = dev.default_device()
one, _, sites = fuzz.synth_pileup(3, genome_len=40000, n_sites=300)
parts, keys = [], []
for i in range(12):
    nm = b"contig_%02d" % i
    parts.append(one.replace(b"synth_chr1", nm))
    if i % 3 != 1: keys += [(nm, p) for _, p in sites]
data = b"".join(parts)
ss = d.siteset(keys, [1] * len(keys))
prm = dev.make_params(0, 0.6, 3, 0, 0.0)
res = d.call_consensus(ss, data, prm, want_counts=False)
d.kernel_timing(True); d.kernel_time_ms(0); d.kernel_time_ms(1)
t = time.time()
for _ in range(5): res = d.call_consensus(ss, data, prm, want_counts=False)
sm, sn = d.kernel_time_ms(0)
print("multi-contig %d bytes: scan kernel %.3f ms (%.0f GB/s), host wall %.1f ms/call, lines %d" % (len(data), sm / sn, len(data) / (sm / sn * 1e-3) / 1e9, (time.time() - t) * 200, res.n_lines))

#!/usr/bin/env python3
"""Development helper: per-wave life times of the scan kernel (SNPGPU_SCAN_MODE=8 SNPGPU_SCAN_DUMP=file), grouped by
XCC / SE / CU / SIMD, to see where waves of equal work run at different speeds."""
import sys

import numpy as np

r = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
r = r[r[:, 0] > 0]
t0 = r[:, 0].min()
start = (r[:, 0] - t0).astype(np.float64) * 0.01
life = (r[:, 1] - r[:, 0]).astype(np.float64) * 0.01
hw = (r[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
xcc = ((r[:, 2] >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
wait, parse, req = (r[:, k].astype(np.float64) for k in (3, 4, 5))
print("waves %d  life us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % (len(life), life.min(), *np.percentile(life, [10, 50, 90]), life.max()))
if len(sys.argv) > 2 and sys.argv[2] == "phases":                # mode 8: rec[6], rec[7] are cycle sums of the terminator scan and the line index
    swar, index = r[:, 6].astype(np.float64), r[:, 7].astype(np.float64)
    tot = wait.sum() + parse.sum() + req.sum() + swar.sum() + index.sum()
    print("cycle shares: dma-wait %.1f%%  terminator scan %.1f%%  line index %.1f%%  line parse + probe %.1f%%  next request %.1f%%"
          % tuple(100 * x / tot for x in (wait.sum(), swar.sum(), index.sum(), parse.sum(), req.sum())))
print("start us: max %.1f | kernel span %.1f us | phase cycles: dma-wait %.3g parse %.3g request %.3g" % (start.max(), (r[:, 1].max() - t0) * 0.01, wait.sum(), parse.sum(), req.sum()))
for name, key in (("xcc", xcc), ("se", se), ("sh", sh), ("cu", cu), ("simd", simd)):
    print(name, " ".join("%d:%.0f(n=%d)" % (k, life[key == k].mean(), (key == k).sum()) for k in np.unique(key)))
phys = xcc * 1000 + se * 100 + sh * 50 + cu
ids, cnt = np.unique(phys, return_counts=True)
print("distinct CUs %d, waves per CU: %s" % (len(ids), dict(zip(*np.unique(cnt, return_counts=True)))))
per_cu = np.array([life[phys == i].mean() for i in ids])
print("per-CU mean life: min %.1f median %.1f max %.1f" % (per_cu.min(), np.median(per_cu), per_cu.max()))
k = np.argsort(life)
for i in list(k[:5]) + list(k[-5:]):
    print("wave %5d xcc %d se %d sh %d cu %2d simd %d start %.1f life %.1f wait %.0f parse %.0f kcycles" % (i, xcc[i], se[i], sh[i], cu[i], simd[i], start[i], life[i], wait[i] / 1e3, parse[i] / 1e3))
simd_key = phys * 10 + simd
ids2, cnt2 = np.unique(simd_key, return_counts=True)
print("waves per SIMD: %s" % dict(zip(*np.unique(cnt2, return_counts=True))))
for c in np.unique(cnt2):
    sel = np.isin(simd_key, ids2[cnt2 == c])
    print("  SIMDs holding %d waves: mean life %.1f" % (c, life[sel].mean()))
n = len(life)
print("life by position in the file (32 bins of the wave index):")
print(" ".join("%.0f" % life[i * n // 32:(i + 1) * n // 32].mean() for i in range(32)))
print("life by wave-in-block:", " ".join("%.0f" % life[w::8].mean() for w in range(8)))
slow = life > 1.5 * np.median(life)
print("slow waves: %d; per block counts of slow waves: %s" % (slow.sum(), dict(zip(*np.unique(slow[:n // 8 * 8].reshape(-1, 8).sum(1), return_counts=True)))))
wpb = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 16
print("life by wave-in-block (%d waves/block):" % wpb, " ".join("%.0f" % life[w::wpb].mean() for w in range(wpb)))
print("simd of wave-in-block:", " ".join("%.1f" % simd[w::wpb].mean() for w in range(wpb)))
# life by age rank on the SIMD (wave-in-block / 4: the host's shares are per rank) — equal lives = balanced shares
rank = (np.arange(len(r)) % 16) // 4 if len(r) % 16 == 0 else None
if rank is not None:
    print("life by age rank:", " ".join("%d:%.1f" % (k, life[rank == k].mean()) for k in range(4)), "| max by rank:", " ".join("%d:%.1f" % (k, life[rank == k].max()) for k in range(4)))

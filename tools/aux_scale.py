#!/usr/bin/env python3
"""Development helper: the small steps (merge_sites, dense windows, region merge, in_regions) at configs[4] scale:
10 000 samples x 2 000 records over 200 000 sites."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snp_pipeline_amd import device as dev
d = dev.default_device()
rng = np.random.default_rng(1)
pos = np.sort(rng.choice(np.arange(1, 5_000_000), size=200_000, replace=False))
n_s, per = 10_000, 2000
keys = np.concatenate([np.sort(rng.choice(pos, size=per, replace=False)) for _ in range(n_s)]).astype(np.uint64)
who = np.repeat(np.arange(n_s, dtype=np.uint32), per)
d.merge_sites(keys[:10], who[:10])
t = time.time(); u, off, car = d.merge_sites(keys, who); t1 = time.time() - t
seg = np.arange(0, n_s * per + 1, per, dtype=np.uint32)
t = time.time(); ws, we, wg = d.dense_windows(keys.astype(np.int64), seg, [3], [1000]); t2 = time.time() - t
t = time.time(); rg, rs, re_ = d.merge_regions(np.zeros(len(ws), np.uint32), ws, we); t3 = time.time() - t
t = time.time(); ins = d.in_regions(np.zeros(len(keys), np.uint32), keys.astype(np.int64), [0, len(rs)], rs, re_); t4 = time.time() - t
print("20 M records: merge_sites %.1f ms (%d unique, %d carriers) | dense %.1f ms (%d) | merge_regions %.1f ms (%d) | in_regions %.1f ms (%d)"
      % (t1 * 1e3, len(u), len(car), t2 * 1e3, len(ws), t3 * 1e3, len(rs), t4 * 1e3, ins.sum()))

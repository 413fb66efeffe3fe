#!/usr/bin/env python3
"""Development helper: snpgpu_merge_sites at BASELINE configs[4] scale (10 000 samples x 1 500 records drawn from 200 000 sites)
against numpy, with wall times.  Usage: python tools/merge_scale.py [n_samples] [records_per_sample] [n_sites]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from snp_pipeline_amd import device as dev
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 200000
    rng = np.random.default_rng(1)
    pool = np.sort(rng.choice(np.arange(1, 5_000_000, dtype=np.uint64), size=S, replace=False)) | (np.uint64(3) << np.uint64(32))
    keys = np.concatenate([rng.choice(pool, size=per, replace=False) for _ in range(n)])
    samp = np.repeat(np.arange(n, dtype=np.uint32), per)
    d = dev.Device(0)
    d.merge_sites(keys[:1000], samp[:1000])
    for rep in range(2):
        t0 = time.time()
        uniq, off, car = d.merge_sites(keys, samp)
        t1 = time.time() - t0
        print("%d records -> %d sites, %d carriers: %.3f s" % (len(keys), len(uniq), len(car), t1))
    t0 = time.time()
    order = np.lexsort((samp, keys))
    ks, ss = keys[order], samp[order]
    wu, first = np.unique(ks, return_index=True)
    t2 = time.time() - t0
    assert np.array_equal(uniq, wu) and np.array_equal(off, np.append(first, len(ks)).astype(np.uint32)) and np.array_equal(car, ss)
    print("equal to numpy (lexsort + unique: %.3f s)" % t2)


if __name__ == "__main__":
    main()

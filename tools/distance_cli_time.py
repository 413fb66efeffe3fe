#!/usr/bin/env python3
"""Development helper: the distance subcommand end to end on a synthetic snpma.fasta (file -> matrix -> kernel -> two TSVs),
wall time per stage.  Usage: python tools/distance_cli_time.py [n_samples] [n_sites]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from snp_pipeline_amd import device as dev
    from snp_pipeline_amd import distance, snp_matrix
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    rng = np.random.default_rng(1)
    tmp = tempfile.mkdtemp(prefix="dist_", dir=os.environ.get("SNPGPU_BENCH_TMP", "/tmp"))
    path = os.path.join(tmp, "snpma.fasta")
    lut = np.frombuffer((b"A" * 24 + b"C" * 24 + b"G" * 24 + b"T" * 24 + b"--NN"), dtype=np.uint8)      # p = .24 x 4, .02, .02
    with open(path, "wb") as f:
        for i in range(n):
            row = lut[rng.integers(0, 100, size=s, dtype=np.uint8)]
            wrapped = np.insert(row, np.arange(60, s, 60), 10)          # 60-column lines, as SeqIO writes consensus.fasta
            f.write(b">SAMPLE%06d\n" % i + wrapped.tobytes() + b"\n")
    d = dev.Device(0)
    for rep in range(2):
        t0 = time.time()
        ids, sym, lens = snp_matrix.load_matrix(path)
        t1 = time.time()
        ids2, mat = distance.distance_of_matrix(d, ids, sym, lens)
        t2 = time.time()
        distance.write_pairwise(os.path.join(tmp, "p.tsv"), ids2, mat)
        t3 = time.time()
        distance.write_matrix(os.path.join(tmp, "m.tsv"), ids2, mat)
        t4 = time.time()
        print("%d x %d (%.0f MB): read %.2f s, pack + distance + copies %.2f s, pairwise TSV %.2f s (%.0f MB), matrix TSV %.2f s; total %.2f s"
              % (n, s, os.path.getsize(path) / 1e6, t1 - t0, t2 - t1, t3 - t2, os.path.getsize(os.path.join(tmp, "p.tsv")) / 1e6, t4 - t3, t4 - t0))
    for name in ("snpma.fasta", "p.tsv", "m.tsv"):
        os.unlink(os.path.join(tmp, name))


if __name__ == "__main__":
    main()

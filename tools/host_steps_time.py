#!/usr/bin/env python3
"""Development helper: wall time of the file-handling subcommands at scale — filter_regions, merge_sites, snp_matrix over N
synthetic sample directories (var.flt.vcf with ~1 500 records each, consensus.fasta with S sites).
Usage: python tools/host_steps_time.py [n_samples] [n_sites]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    from snp_pipeline_amd import varscan
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    G = 5_000_000
    rng = np.random.default_rng(3)
    tmp = tempfile.mkdtemp(prefix="hoststeps_", dir=os.environ.get("SNPGPU_BENCH_TMP", "/tmp"))
    try:
        ref = os.path.join(tmp, "ref.fasta")
        with open(ref, "w") as f:
            f.write(">synth_chr1\n" + "\n".join("A" * 60 for _ in range(G // 60)) + "\n")
        site_pool = np.sort(rng.choice(np.arange(600, G - 600), size=S, replace=False))
        header = varscan.header_text(15)
        row = "synth_chr1\t%d\t.\tG\tA\t.\tPASS\tADP=26;WT=0;HET=0;HOM=1;NC=0\t" + varscan.FORMAT_KEYS + "\t1/1:146:28:26:0:26:100%%:2.0165E-15:0:30:0:0:13:13\n"
        letters = np.frombuffer(b"ACGT-", dtype=np.uint8)
        dirs = []
        for i in range(n):
            d = os.path.join(tmp, "samples", "S%05d" % i)
            os.makedirs(d)
            pos = np.sort(rng.choice(site_pool, size=1500, replace=False))
            with open(os.path.join(d, "var.flt.vcf"), "w") as f:
                f.write(header + "".join(row % p for p in pos))
            seq = rng.choice(letters, size=S).tobytes().decode()
            with open(os.path.join(d, "consensus.fasta"), "w") as f:
                f.write(">S%05d\n" % i + "\n".join(seq[k:k + 60] for k in range(0, S, 60)) + "\n")
            dirs.append(d)
        dirs_file = os.path.join(tmp, "sampleDirectories.txt")
        with open(dirs_file, "w") as f:
            f.write("\n".join(dirs) + "\n")

        def run(line):
            args = cli.parse_command_line(line)
            args.verbose = 0
            t0 = time.time()
            cli.run_command_from_args(args)
            return time.time() - t0

        run("merge_sites -n var.flt.vcf -o %s/warm.txt %s %s.warm" % (tmp, dirs_file, dirs_file))      # library load, device context
        t_f = run("filter_regions -n var.flt.vcf %s %s --edge_length 500 --window_size 1000 125 15 --max_snp 3 2 1" % (dirs_file, ref))
        t_m = run("merge_sites -f -n var.flt.vcf -o %s/snplist.txt %s %s.filtered" % (tmp, dirs_file, dirs_file))
        t_s = run("snp_matrix -c consensus.fasta -o %s/snpma.fasta %s.filtered" % (tmp, dirs_file))
        print("%d samples: filter_regions %.2f s, merge_sites %.2f s (snplist %.0f MB), snp_matrix %.2f s (snpma %.0f MB)"
              % (n, t_f, t_m, os.path.getsize(os.path.join(tmp, "snplist.txt")) / 1e6, t_s, os.path.getsize(os.path.join(tmp, "snpma.fasta")) / 1e6))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()

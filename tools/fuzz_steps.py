#!/usr/bin/env python3
"""Development helper: a timed campaign of the separate subcommands, chained the way run.py chains them, against the chain of
CPU restatements (varscan_oracle -> steps_oracle -> pileup_oracle -> steps_oracle) on FRESH seeds: a small synthetic outbreak of
random shape, random options per step.  tests/test_gpu_pipeline.py does this for one seed; tools/fuzz_jobs.py compares the one
job with these steps.  Usage: python tools/fuzz_steps.py [seconds] [first seed]; the first disagreement ends the run (exit 1)."""
import os
import pathlib
import random
import shutil
import sys
import tempfile
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) % 1000000
    sys.argv = ["cfsan_snp_pipeline", "fuzz_steps"]
    from oracle import fuzz
    from oracle import pileup_oracle as po
    from oracle import steps_oracle as so
    from oracle import varscan_oracle as vo
    from tests import test_gpu_pipeline as tp
    run, fasta = tp._run, tp._fasta
    home = os.getcwd()
    done, seed = 0, seed0
    t_end = time.time() + seconds
    while time.time() < t_end:
        seed += 1
        rng = random.Random(seed)
        work = pathlib.Path(tempfile.mkdtemp(prefix="steps_%d_" % seed))
        what = {}
        try:
            n = rng.choice([1, 2, 4, 6])
            glen = rng.choice([2500, 6000, 12000])
            contigs = rng.choice([("ctg1", "ctg2"), ("only",), ("b_contig", "a_contig", "NODE_3_length_%d_cov_2.5" % glen)])
            refs, piles = fuzz.cohort_pileups(seed, n_samples=n, genome_len=glen, contigs=contigs, mean_depth=rng.choice([12, 22, 40]), n_scattered=rng.choice([4, 9, 30]))
            for i in range(n):
                variant = rng.choice([None, None, None, "crlf", "mixed"])
                if variant:
                    piles[i] = fuzz.with_line_ends(piles[i], variant, seed + i)
            names = ["iso%02d" % i for i in range(n)]
            ref_path = work / "reference" / "ref.fasta"
            ref_path.parent.mkdir()
            ref_path.write_text("".join(fasta(c, refs[c]) for c in refs))
            old = time.time() - 1000
            os.utime(str(ref_path), (old, old))
            dirs = []
            for name, data in zip(names, piles):
                sdir = work / "samples" / name
                sdir.mkdir(parents=True)
                bam = sdir / "reads.sorted.deduped.indelrealigned.bam"
                bam.write_bytes(b"placeholder")
                os.utime(str(bam), (old, old))
                (sdir / "reads.all.pileup").write_bytes(data)
                dirs.append(str(sdir))
            dirs_file = str(work / "sampleDirectories.txt")
            order = list(dirs)
            rng.shuffle(order)
            with open(dirs_file, "w") as f:
                f.write("\n".join(order) + "\n")
            vs_extra, vs_kw = rng.choice([("--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5", dict(vo.PIPELINE_DEFAULTS)),
                                          ("--min-avg-qual 0 --min-var-freq 0.3 --min-freq-for-hom 0.95", dict(min_avg_qual=0, min_var_freq=0.3, min_freq_for_hom=0.95)),
                                          ("--min-var-freq 0.5 --min-reads2 3 --p-value 1e-6 --strand-filter 0 --min-coverage 10",
                                           dict(min_var_freq=0.5, min_reads2=3, p_value=1e-6, strand_filter=0, min_coverage=10))])
            mode = rng.choice(["all", "each"])
            edge = rng.choice([1, 100, 500, 5000])
            windows, max_snps = rng.choice([([1000, 125, 15], [3, 2, 1]), ([500], [2]), ([60, 2000], [1, 6])])
            outgroup = set(rng.sample(names, 1)) if (n > 1 and rng.random() < 0.3) else set()
            max_n = rng.choice([-1, -1, 10, 25, 1000])
            q, c, D, d, b = rng.choice([(15, 0.9, 5, 2, 0.1), (0, 0.6, 3, 0, 0.0), (30, 0.75, 2, 1, 0.25), (10, 0.51, 1, 0, 0.0)])
            what = dict(seed=seed, n=n, glen=glen, contigs=contigs, varscan=vs_extra, mode=mode, edge=edge, windows=windows, max_snps=max_snps,
                        outgroup=sorted(outgroup), maxsnps=max_n, caller=(q, c, D, d, b))
            os.environ["VarscanMpileup2snp_ExtraParams"] = vs_extra
            os.environ.pop("errorOutputFile", None)
            os.chdir(str(work))
            # call_sites
            sites = {}
            for name, data, sdir in zip(names, piles, dirs):
                run("call_sites %s %s" % (ref_path, sdir))
                want = vo.mpileup2snp(data, vo.Params(**vs_kw))
                assert open(os.path.join(sdir, "var.flt.vcf")).read() == want, ("var.flt.vcf", name)
                sites[name] = [(ln.split("\t")[0], int(ln.split("\t")[1])) for ln in want.splitlines() if not ln.startswith("#")]
            # filter_regions
            lens = {cn: len(refs[cn]) for cn in refs}
            og = ""
            if outgroup:
                (work / "outgroup.txt").write_text("".join(x + "\n" for x in sorted(outgroup)))
                og = " --out_group %s/outgroup.txt" % work
            run("filter_regions -n var.flt.vcf %s %s --edge_length %d --window_size %s --max_snp %s --mode %s%s"
                % (dirs_file, ref_path, edge, " ".join(map(str, windows)), " ".join(map(str, max_snps)), mode, og))
            bad = so.bad_regions([(nm, sites[nm]) for nm in names], lens, edge, max_snps, windows, mode=mode, outgroup=outgroup)
            kept, removed = {}, {}
            for name, sdir in zip(names, dirs):
                mine = {} if name in outgroup else (bad if mode == "all" else bad[name])
                kept[name] = [k for k in sites[name] if not so.in_region(k[1], mine.get(k[0], []))]
                removed[name] = [k for k in sites[name] if so.in_region(k[1], mine.get(k[0], []))]
                src = [ln for ln in open(os.path.join(sdir, "var.flt.vcf")).read().splitlines(True) if not ln.startswith("#")]
                for fname, keys in (("var.flt_preserved.vcf", kept[name]), ("var.flt_removed.vcf", removed[name])):
                    got = [ln for ln in open(os.path.join(sdir, fname)).read().splitlines(True) if not ln.startswith("#")]
                    keyset = set(keys)
                    assert got == [ln for ln in src if (ln.split("\t")[0], int(ln.split("\t")[1])) in keyset], (fname, name)
            # merge_sites, both flows; then per flow: call_consensus, snp_matrix, distance, snp_reference
            for suffix, vcf, per_sample in (("", "var.flt.vcf", sites), ("_preserved", "var.flt_preserved.vcf", kept)):
                snplist = str(work / ("snplist%s.txt" % suffix))
                run("merge_sites -n %s --maxsnps %d -o %s %s %s.filtered%s" % (vcf, max_n, snplist, dirs_file, dirs_file, suffix))
                merged, excluded = so.merge_sites([(dd, nm, per_sample[nm]) for dd, nm in sorted(zip(dirs, names))], max_n)
                assert open(snplist).read() == so.snplist_text(merged), ("snplist", suffix)
                assert open(dirs_file + ".filtered" + suffix).read() == "".join(dd + "\n" for dd in order if dd not in excluded), ("filtered list", suffix)
                snp_keys = [(k[0].encode(), k[1]) for k, _ in merged]
                seqs = {}
                for name, data, sdir in zip(names, piles, dirs):
                    excl = " -e %s/var.flt_removed.vcf" % sdir if suffix else ""
                    run("call_consensus -l %s%s -o %s/consensus%s.fasta -q %d -c %s -D %d -d %d -b %s %s/reads.all.pileup" % (snplist, excl, sdir, suffix, q, c, D, d, b, sdir))
                    excluded_keys = set((cn.encode(), p) for cn, p in removed[name]) if suffix else set()
                    want, _ = po.call_consensus_sites(data, snp_keys, excluded_keys, po.CallerParams(q, c, D, d, b))
                    seqs[name] = want.decode()
                    assert open(os.path.join(sdir, "consensus%s.fasta" % suffix)).read() == fasta(name, seqs[name]), ("consensus", suffix, name)
                members = [nm for dd, nm in sorted(zip(dirs, names)) if dd not in excluded]
                if not members:
                    continue                                     # (snp_matrix ends the run with a global error: regression scenarios)
                snpma = str(work / ("snpma%s.fasta" % suffix))
                run("snp_matrix -c consensus%s.fasta -o %s %s.filtered%s" % (suffix, snpma, dirs_file, suffix))
                assert open(snpma).read() == "".join(fasta(nm, seqs[nm]) for nm in members), ("snpma", suffix)
                run("distance -p %s/pairs%s.tsv -m %s/matrix%s.tsv %s" % (work, suffix, work, suffix, snpma))
                ids, table = so.distance_tables({nm: seqs[nm] for nm in members})
                assert open(str(work / ("pairs%s.tsv" % suffix))).read() == so.pairwise_text(ids, table), ("pairs", suffix)
                assert open(str(work / ("matrix%s.tsv" % suffix))).read() == so.matrix_text(ids, table), ("matrix", suffix)
                run("snp_reference -l %s -o %s/referenceSNP%s.fasta %s" % (snplist, work, suffix, ref_path))
                want_ref = ""
                for cn in sorted(refs):
                    bases = "".join(refs[cn][p - 1].upper() for k, p in [kk for kk, _ in merged] if k == cn)
                    want_ref += fasta(cn, bases)                 # (a record for every contig of the reference, with or without positions: utils.py:1103-1110)
                assert open(str(work / ("referenceSNP%s.fasta" % suffix))).read() == want_ref, ("referenceSNP", suffix)
            done += 1
        except BaseException:                                    # noqa: B902
            print("DISAGREEMENT %r\n%s" % (what, traceback.format_exc()[-1500:]))
            print("chains that agreed before that: %d" % done)
            sys.exit(1)
        finally:
            os.chdir(home)
            shutil.rmtree(str(work), ignore_errors=True)
    print("fuzz steps: %.0f s, seeds %d..%d, %d chains of subcommands equal to the chain of restatements" % (seconds, seed0 + 1, seed, done))


if __name__ == "__main__":
    main()

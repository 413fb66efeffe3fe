#!/usr/bin/env python3
"""Development helper: a timed campaign of whole jobs on FRESH seeds — a small synthetic outbreak with random shape and random
step options through the separate subcommands (what run.py starts) and through ONE hot_path_batch job; every output file must
be the same bytes.  Usage: python tools/fuzz_jobs.py [seconds] [first seed, 0 = from the clock] [n: every n-th job also sharded over 2 / 3 ranks]; the first disagreement is kept under
gpurun_out/fuzz_jobs/ (the options and the names of the files that differ) and ends the run with exit code 1."""
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
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) else int(time.time()) % 1000000
    argv = list(sys.argv)
    from oracle import fuzz
    from tests import test_gpu_pipeline as tp
    out_dir = os.path.join(ROOT, "gpurun_out", "fuzz_jobs")
    home = os.getcwd()
    done, stopped, sharded, failing, seed = 0, 0, 0, 0, seed0
    existing = 0
    ranks_every = int(argv[3]) if len(argv) > 3 else 0             # every n-th job also sharded over 2 or 3 ranks
    sys.argv = ["cfsan_snp_pipeline", "fuzz_jobs"]
    t_end = time.time() + seconds
    while time.time() < t_end:
        seed += 1
        rng = random.Random(seed)
        work = pathlib.Path(tempfile.mkdtemp(prefix="job_%d_" % seed))
        what = {}
        try:
            n = rng.choice([1, 2, 3, 5, 7])
            what["tree"] = dict(seed=seed, n_samples=n, genome_len=rng.choice([2500, 6000, 12000]))
            ref_path, dirs, dirs_file, piles = tp._outbreak_tree(work, **what["tree"])
            # some samples with other line ends, one now and then with positions that come twice
            for i, sdir in enumerate(dirs):
                variant = rng.choice([None, None, None, "crlf", "mixed", "repeats"])
                if variant:
                    with open(os.path.join(sdir, "reads.all.pileup"), "wb") as f:
                        f.write(fuzz.with_line_ends(piles[i], variant, seed + i))
                    what.setdefault("line_ends", {})[os.path.basename(sdir)] = variant
            mode = rng.choice(["all", "each"])
            rules = rng.choice([("1000 125 15", "3 2 1"), ("500", "2"), ("60 2000", "1 6")])
            filter_extra = "--edge_length %d --window_size %s --max_snp %s --mode %s" % (rng.choice([1, 100, 500, 5000]), rules[0], rules[1], mode)
            if n > 1 and rng.random() < 0.3:
                og = work / "outgroup.txt"
                og.write_text("".join(os.path.basename(d) + "\n" for d in rng.sample(dirs, rng.choice([1, 1, 2][:n - 1] or [1]))))
                filter_extra += " --out_group %s" % og
            merge_extra = rng.choice(["", "", "--maxsnps %d" % rng.choice([0, 10, 25, 40, 1000])])
            consensus_extra = rng.choice([tp.CONSENSUS_EXTRA, "-q 0 -c 0.6 -D 3 -d 0 -b 0", "-q 30 -c 0.75 -D 2 -d 1 -b 0.25 --vcfFailedSnpGt 1",
                                          "-q 15 -c 0.9 -D 5 -d 2 -b 0.1 --vcfPreserveRefCase", "-q 10 -c 0.51 -D 1 -d 0 -b 0.0",
                                          "-q 10 -c 0.51 -D 1 -d 0 -b 0.0 --vcfAllPos"])
            varscan_extra = rng.choice([tp.VARSCAN_EXTRA, "--min-avg-qual 0 --min-var-freq 0.3 --min-reads2 2", "--min-var-freq 0.5 --min-reads2 3 --p-value 1e-6 --strand-filter 0"])
            if rng.random() < 0.25:                              # collect_metrics by-products: the samples' metrics files are compared too
                consensus_extra += " --amdMetricsRefFasta %s" % ref_path
            resident = int(rng.choice([1.2, 2.5]) * max(len(p) for p in piles)) if (n > 2 and rng.random() < 0.25) else 0   # part of the pileups streamed twice
            what.update(filter=filter_extra, merge=merge_extra, consensus=consensus_extra, varscan=varscan_extra, resident=resident)

            def metrics():
                return {os.path.basename(d) + "/metrics": open(os.path.join(d, "metrics"), "rb").read() for d in dirs if os.path.exists(os.path.join(d, "metrics"))}
            os.environ["VarscanMpileup2snp_ExtraParams"] = varscan_extra
            os.environ.pop("errorOutputFile", None)
            os.chdir(str(work))
            if n >= 2 and rng.random() < 0.2:
                # one sample whose pileup has a position the reference cannot convert (site calling takes the column as text, call_consensus
                # ends with ValueError): StopOnSampleError=false, the others go on — in the separate steps and in the job alike
                from snp_pipeline_amd import cfsan_snp_pipeline as cli
                bad_dir = rng.choice(dirs)
                bad_path = os.path.join(bad_dir, "reads.all.pileup")
                lines = open(bad_path, "rb").read().split(b"\n")
                k = rng.randrange(len(lines) - 1)
                f = lines[k].split(b"\t")
                f[1] = f[1] + b"x"
                lines[k] = b"\t".join(f)
                open(bad_path, "wb").write(b"\n".join(lines))
                what["failing"] = os.path.basename(bad_dir)
                os.environ["StopOnSampleError"] = "false"
                os.environ["errorOutputFile"] = str(work / "error.log")
                good = [d for d in dirs if d != bad_dir]

                def run(line):
                    a = cli.parse_argument_list([w.replace("\x00", " ") for w in line.split()])
                    a.verbose = 0
                    try:
                        return cli.run_command_from_args(a)
                    except SystemExit as e:
                        if e.code != 98:                         # a global error (every sample excluded ...): the caller counts it as stopped
                            raise
                        return 98
                    except (ValueError, IndexError):
                        return 98
                try:
                    for sdir in dirs:
                        run("call_sites %s %s" % (ref_path, sdir))
                    run("filter_regions -f -n var.flt.vcf %s %s %s" % (dirs_file, ref_path, filter_extra))
                    run("merge_sites -f -n var.flt.vcf -o %s/snplist.txt %s %s %s.OrigVCF.filtered" % (work, merge_extra, dirs_file, dirs_file))
                    run("merge_sites -f -n var.flt_preserved.vcf -o %s/snplist_preserved.txt %s %s %s.PresVCF.filtered" % (work, merge_extra, dirs_file, dirs_file))
                    for sdir in dirs:
                        run("call_consensus -f -l %s/snplist.txt -o %s/consensus.fasta --vcfRefName ref.fasta %s --vcfFileName consensus.vcf %s/reads.all.pileup"
                            % (work, sdir, consensus_extra, sdir))
                        run("call_consensus -f -l %s/snplist_preserved.txt -o %s/consensus_preserved.fasta -e %s/var.flt_removed.vcf --vcfRefName ref.fasta %s "
                            "--vcfFileName consensus_preserved.vcf %s/reads.all.pileup" % (work, sdir, sdir, consensus_extra, sdir))
                    for suffix, flt in (("", "OrigVCF"), ("_preserved", "PresVCF")):
                        run("snp_matrix -f -c consensus%s.fasta -o %s/snpma%s.fasta %s.%s.filtered" % (suffix, work, suffix, dirs_file, flt))
                        run("snp_reference -f -l %s/snplist%s.txt -o %s/referenceSNP%s.fasta %s" % (work, suffix, work, suffix, ref_path))
                        run("distance -f -p %s/snp_distance_pairwise%s.tsv -m %s/snp_distance_matrix%s.tsv %s/snpma%s.fasta" % (work, suffix, work, suffix, work, suffix))
                    var_files = ("var.flt.vcf", "var.flt_preserved.vcf", "var.flt_removed.vcf")
                    want = tp._snapshot(work, good)
                    want.update({"bad/" + nm: open(os.path.join(bad_dir, nm), "rb").read() for nm in var_files})
                    left = sorted(nm for nm in tp.PER_SAMPLE if os.path.exists(os.path.join(bad_dir, nm)))
                    for nm in left:
                        os.remove(os.path.join(bad_dir, nm))
                    if os.path.exists(os.path.join(bad_dir, "metrics")):
                        os.remove(os.path.join(bad_dir, "metrics"))
                    for d in good:
                        if os.path.exists(os.path.join(d, "metrics")):
                            os.remove(os.path.join(d, "metrics"))
                    rc = run("hot_path_batch -f %s %s --filterRegionsExtraParams=%s --mergeSitesExtraParams=%s --callConsensusExtraParams=%s"
                             % (dirs_file, ref_path, filter_extra.replace(" ", "\x00"), merge_extra.replace(" ", "\x00"), consensus_extra.replace(" ", "\x00")))
                    assert rc in (0, 98)
                    got = tp._snapshot(work, good, remove=False)
                    got.update({"bad/" + nm: open(os.path.join(bad_dir, nm), "rb").read() for nm in var_files})
                    differ = [kk for kk in sorted(want) if got.get(kk) != want[kk]]
                    assert not differ, "with a failing sample, files that differ: %r" % differ
                    assert sorted(nm for nm in tp.PER_SAMPLE if os.path.exists(os.path.join(bad_dir, nm))) == left, "the failing sample's files"
                    failing += 1
                except SystemExit as stop:                       # a global error of a later step (e.g. every sample excluded): not this branch's subject
                    if stop.code not in (100,):
                        raise
                    stopped += 1
                finally:
                    os.environ.pop("StopOnSampleError", None)
                    os.environ.pop("errorOutputFile", None)
                continue
            job = ("hot_path_batch -f %s %s --filterRegionsExtraParams=%s --mergeSitesExtraParams=%s --callConsensusExtraParams=%s"
                   % (dirs_file, ref_path, filter_extra.replace(" ", "\x00"), merge_extra.replace(" ", "\x00"), consensus_extra.replace(" ", "\x00"))) + (" --residentBytes %d" % resident if resident else "")
            try:
                tp._separate_steps(work, ref_path, dirs, dirs_file, filter_extra, merge_extra, consensus_extra)
            except SystemExit as stop:                           # e.g. --maxsnps took every sample out: snp_matrix ends the run (exit 100)
                try:
                    tp._run(job)
                except SystemExit as stop2:
                    assert stop2.code == stop.code, "the steps stopped with %r, the one job with %r" % (stop.code, stop2.code)
                    stopped += 1
                    continue
                raise AssertionError("the separate steps stopped with exit code %r, the one job went through" % (stop.code,))
            want = tp._snapshot(work, dirs)
            want.update(metrics())
            for d in dirs:
                if os.path.exists(os.path.join(d, "metrics")):
                    os.remove(os.path.join(d, "metrics"))
            # the batch subcommands (every sample of the list in one process) write what the per-sample commands wrote
            tp._run("call_sites_batch %s %s" % (ref_path, dirs_file))
            for name in ("snplist.txt", "snplist_preserved.txt"):
                with open(os.path.join(str(work), name), "wb") as f:
                    f.write(want[name])
            for d in dirs:
                with open(os.path.join(d, "var.flt_removed.vcf"), "wb") as f:
                    f.write(want[os.path.join(os.path.basename(d), "var.flt_removed.vcf")])
            tp._run("call_consensus_batch -f -l %s/snplist.txt -o consensus.fasta --vcfRefName ref.fasta %s --vcfFileName consensus.vcf %s" % (work, consensus_extra, dirs_file))
            tp._run("call_consensus_batch -f -l %s/snplist_preserved.txt -e var.flt_removed.vcf -o consensus_preserved.fasta --vcfRefName ref.fasta %s "
                    "--vcfFileName consensus_preserved.vcf %s" % (work, consensus_extra, dirs_file))
            batch_differ = []
            for d in dirs:
                for name in ("var.flt.vcf", "consensus.fasta", "consensus.vcf", "consensus_preserved.fasta", "consensus_preserved.vcf"):
                    key = os.path.join(os.path.basename(d), name)
                    if open(os.path.join(d, name), "rb").read() != want[key]:
                        batch_differ.append("batch:" + key)
                    os.remove(os.path.join(d, name))
                os.remove(os.path.join(d, "var.flt_removed.vcf"))
                if os.path.exists(os.path.join(d, "metrics")):
                    if open(os.path.join(d, "metrics"), "rb").read() != want.get(os.path.join(os.path.basename(d), "metrics")):
                        batch_differ.append("batch:" + os.path.basename(d) + "/metrics")
                    os.remove(os.path.join(d, "metrics"))
            for name in ("snplist.txt", "snplist_preserved.txt"):
                os.remove(os.path.join(str(work), name))
            assert not batch_differ, "the batch subcommands' files differ: %r" % batch_differ
            tp._run(job)
            got = tp._snapshot(work, dirs, remove=False)
            got.update(metrics())
            differ = [k for k in sorted(want) if got.get(k) != want[k]] + [k for k in got if k not in want]
            if not differ and done % 3 == 1:
                # the same job with the var.flt.vcf files as INPUTS (--siteCalling existing): they keep bytes and modification
                # times, every other file comes out the same
                stamps = [os.stat(os.path.join(d, "var.flt.vcf")).st_mtime_ns for d in dirs]
                tp._run(job + " --siteCalling existing")
                again = tp._snapshot(work, dirs, remove=False)
                again.update(metrics())
                differ = ["existing:" + k for k in sorted(got) if again.get(k) != got[k]]
                if stamps != [os.stat(os.path.join(d, "var.flt.vcf")).st_mtime_ns for d in dirs]:
                    differ.append("existing: a var.flt.vcf was rewritten")
                existing += 1
            if not differ and ranks_every and done % ranks_every == 0:
                # the same job sharded over 2 or 3 ranks (torchrun; all ranks on this one GPU, gloo between them): the same files again
                import socket
                import subprocess
                world = rng.choice([2, 3])
                what["world"] = world
                for d in dirs:
                    for name in list(tp.PER_SAMPLE) + ["metrics"]:
                        if os.path.exists(os.path.join(d, name)):
                            os.remove(os.path.join(d, name))
                for name in tp.TOP_LEVEL:
                    os.remove(os.path.join(str(work), name))
                s = socket.socket()
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
                s.close()
                env = dict(os.environ, SNPGPU_PIPELINE_ONE_GPU="1", MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
                cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                       "--master-port", str(port), os.path.join(ROOT, "bin", "cfsan_snp_pipeline")] + [w.replace("\x00", " ") for w in job.split()] + ["-v", "0"]
                r = subprocess.run(cmd, cwd=str(work), env=env, capture_output=True, text=True, timeout=900)
                assert r.returncode == 0, "sharded job failed: %s" % (r.stdout[-1500:] + r.stderr[-3000:])
                again = tp._snapshot(work, dirs, remove=False)
                again.update(metrics())
                differ = ["ranks%d:" % world + k for k in sorted(got) if again.get(k) != got[k]]
                want, got = got, again
                sharded += 1
            if differ:                                           # keep both versions of the first few, and the small top-level inputs
                keep = os.path.join(out_dir, "job_%d" % seed)
                os.makedirs(keep, exist_ok=True)
                for k in [x.split(":", 1)[-1] for x in differ[:4]] + [x for x in ("snplist.txt", "snplist_preserved.txt", "sampleDirectories.txt.OrigVCF.filtered", "sampleDirectories.txt.PresVCF.filtered") if x in want]:
                    for tag, src in (("steps", want), ("job", got)):
                        with open(os.path.join(keep, k.replace("/", "_") + "." + tag), "wb") as f:
                            f.write(src.get(k, b""))
                for sdir in dirs:
                    shutil.copy(os.path.join(sdir, "var.flt_removed.vcf"), os.path.join(keep, os.path.basename(sdir) + "_var.flt_removed.vcf"))
            assert not differ, "files that differ: %r" % differ
            done += 1
        except BaseException:                                    # noqa: B902 — SystemExit of a step included: that is a finding too
            os.makedirs(out_dir, exist_ok=True)
            with open(os.path.join(out_dir, "job_%d.txt" % seed), "w") as f:
                f.write(repr(what) + "\n" + traceback.format_exc())
            print("DISAGREEMENT seed=%d %r\n%s" % (seed, what, traceback.format_exc().splitlines()[-1]))
            print("jobs that agreed before that: %d" % done)
            sys.exit(1)
        finally:
            os.chdir(home)
            shutil.rmtree(str(work), ignore_errors=True)
    print("fuzz jobs: %.0f s, seeds %d..%d, %d jobs with every output file of the one job equal to the separate steps', %d that both ways stopped alike"
          % (seconds, seed0 + 1, seed, done, stopped) + (", %d of them again sharded over 2 / 3 ranks" % sharded if ranks_every else "")
          + ", %d jobs with a failing sample alike" % failing + ", %d again with --siteCalling existing" % existing)


if __name__ == "__main__":
    main()

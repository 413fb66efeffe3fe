#!/usr/bin/env python3
"""Generate the golden vectors under ``tests/golden/`` by running the REAL
reference (``/root/reference/snppipeline``) in the build container.

Run from the repo root:  ``python oracle/gen_golden.py``

The reference is imported in place (never copied, no bytecode written).
``snppipeline.pileup`` imports natively; ``utils``/``filter_regions``/
``call_consensus`` need ``Bio`` and ``vcf`` (PyVCF3), which are not installed,
so minimal stand-in modules are registered in ``sys.modules`` — enough for the
functions exercised here to run their own arithmetic:
``vcf.Reader`` yields objects with ``CHROM``/``POS`` parsed from the data lines
(all the reference ever reads from it on these paths), ``SeqIO.write`` captures
the sequence string that ``call_consensus`` built.

Outputs (committed):
  tests/golden/pileup_vectors.json.gz   strip / Record / caller vectors, whole-file runs
  tests/golden/steps_vectors.json.gz    region, merge, distance vectors
  tests/golden/fixtures/...             the reference's bundled ExpectedResults files
                                        that pin the path (data files, some gzipped)
"""

import argparse
import doctest
import gzip
import io
import json
import os
import random
import shutil
import sys
import tarfile
import tempfile
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")

sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)


def install_stubs():
    captured = {}

    bio = types.ModuleType("Bio")
    seqio = types.ModuleType("Bio.SeqIO")
    seqm = types.ModuleType("Bio.Seq")
    recm = types.ModuleType("Bio.SeqRecord")

    class Seq(str):
        pass

    class SeqRecord(object):
        def __init__(self, seq, id="", description=""):
            self.seq, self.id, self.description = seq, id, description

        def __len__(self):
            return len(self.seq)

    def write(records, handle, fmt):
        for r in records:
            captured["last"] = (r.id, str(r.seq))
            handle.write(">%s\n" % r.id)
        return len(records)

    def parse(handle, fmt):
        if isinstance(handle, str):                     # SeqIO.parse also takes a path (collect_metrics.py:334)
            with open(handle) as f:
                for rec in parse(f, fmt):
                    yield rec
            return
        name, chunks = None, []
        for line in handle:
            if line.startswith(">"):
                if name is not None:
                    yield SeqRecord(Seq("".join(chunks)), id=name)
                name, chunks = line[1:].split()[0], []
            else:
                chunks.append(line.strip())
        if name is not None:
            yield SeqRecord(Seq("".join(chunks)), id=name)

    seqio.write, seqio.parse = write, parse
    seqm.Seq, recm.SeqRecord = Seq, SeqRecord
    bio.SeqIO, bio.Seq, bio.SeqRecord = seqio, seqm, recm

    vcf = types.ModuleType("vcf")

    class _Rec(object):
        def __init__(self, chrom, pos):
            self.CHROM, self.POS = chrom, pos

    class Reader(object):
        def __init__(self, fsock=None, **kw):
            self._f = fsock

        def __iter__(self):
            for line in self._f:
                if line.startswith("#") or not line.strip():
                    continue
                f = line.split("\t")
                yield _Rec(f[0], int(f[1]))

    class Writer(object):
        """Stand-in for vcf.Writer: the decisions (which record goes to which file) as "CHROM<TAB>POS" lines."""
        def __init__(self, stream, template=None, **kw):
            self._s = stream

        def write_record(self, rec):
            self._s.write("%s\t%d\n" % (rec.CHROM, rec.POS))

        def flush(self):
            self._s.flush()

        def close(self):
            self._s.close()

    vcf.Reader = Reader
    vcf.Writer = Writer
    for name, mod in [("Bio", bio), ("Bio.SeqIO", seqio), ("Bio.Seq", seqm), ("Bio.SeqRecord", recm), ("vcf", vcf)]:
        sys.modules[name] = mod
    return captured


def counter_items(c):
    return sorted([k, v] for k, v in c.items())


PARAM_SETS = [
    [0, 0.6, 1, 0, 0.0],      # CLI defaults (cfsan_snp_pipeline.py:397-401)
    [0, 0.6, 3, 0, 0.0],      # pipeline conf (snppipeline.conf:249)
    [15, 0.9, 5, 2, 0.1],     # strict set from SURVEY 8(d)
    [20, 0.75, 2, 1, 0.25],
    [0, 1.0, 0, 0, 0.5],
]


def record_vector(pileup, line, with_calls=True):
    """Run the reference Record (+ callers) on one line for every parameter set."""
    out = {"line": line, "by_q": {}}
    for q in sorted({p[0] for p in PARAM_SETS}):
        try:
            r = pileup.Record(line, q)
        except Exception as e:                      # IndexError / ValueError paths
            out["by_q"][str(q)] = {"error": type(e).__name__}
            continue
        d = {
            "chrom": r.chrom, "pos": r.position, "ref": r.reference_base, "raw": r.raw_depth,
            "good": r.good_depth, "fwd": r.forward_good_depth, "rev": r.reverse_good_depth,
            "total_hist": counter_items(r.base_good_depth),
            "fwd_hist": counter_items(r.forward_base_good_depth),
            "rev_hist": counter_items(r.reverse_base_good_depth),
            "ranked": r.most_common_good_bases,
            "calls": [],
        }
        if with_calls:
            for p in PARAM_SETS:
                if p[0] != q:
                    continue
                caller = pileup.ConsensusCaller(p[1], p[2], p[3], p[4])
                base, failed = caller.call_consensus(r)
                d["calls"].append({"params": p, "base": base, "failed": failed})
        out["by_q"][str(q)] = d
    return out


def gen_file_runs(captured, specs, extra_sites=(), line_ends=None):
    """Whole files through the reference's own call_consensus driver.  specs: (seed, synth_pileup kwargs, params).
    line_ends: fuzz.with_line_ends variant applied to the synthetic file (recorded in the run)."""
    # --- whole-file runs through the reference's own call_consensus driver
    from oracle import fuzz
    from snppipeline import call_consensus as cc
    from snppipeline import utils as ref_utils
    runs = []
    tmp = tempfile.mkdtemp(prefix="golden_")
    try:
        for seed, kw, pset in specs:
            data, refs, sites = fuzz.synth_pileup(seed, **kw)
            if line_ends:
                data = fuzz.with_line_ends(data, line_ends, seed)
            rng = random.Random(seed)
            # snplist: sites (+ some positions that have no pileup line, + one duplicate)
            snps = list(sites)
            snps.append((sites[0][0], 10_000_000))
            if extra_sites:
                snps = sorted(set(snps + [(sites[0][0], p) for p in extra_sites]))
            snps.sort()
            excluded = sorted(rng.sample(sites, max(1, len(sites) // 7)))
            extra_excl = [(sites[0][0], 5), (sites[0][0], 6)]       # excluded but not in snplist
            sdir = os.path.join(tmp, "sample%d" % seed)
            os.makedirs(sdir)
            ppath = os.path.join(sdir, "reads.all.pileup")
            with open(ppath, "wb") as f:
                f.write(data)
            lpath = os.path.join(tmp, "snplist%d.txt" % seed)
            with open(lpath, "w") as f:
                for c, p in snps:
                    f.write("%s\t%d\t1\tx\n" % (c.decode(), p))
            epath = os.path.join(sdir, "excl.vcf")
            with open(epath, "w") as f:
                f.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n")
                for c, p in excluded + extra_excl:
                    f.write("%s\t%d\t.\tA\tC\t.\tPASS\t.\tGT\t1/1\n" % (c.decode(), p))
            for use_excl in (False, True):
                args = argparse.Namespace(
                    snpListFile=lpath, allPileupFile=ppath, consensusFile=os.path.join(sdir, "consensus.fasta"),
                    excludeFile=epath if use_excl else None, forceFlag=True, vcfFileName=None, vcfRefName="x",
                    vcfAllPos=False, vcfPreserveRefCase=False, vcfFailedSnpGt=".", minBaseQual=pset[0],
                    minConsFreq=pset[1], minConsDpth=pset[2], minConsStrdDpth=pset[3], minConsStrdBias=pset[4])
                ref_utils.log_verbosity = 0
                sink = io.StringIO()
                old = sys.stdout
                sys.stdout = sink
                try:
                    cc.call_consensus(args)
                finally:
                    sys.stdout = old
                sid, seq = captured["last"]
                runs.append({
                    "seed": seed, "kw": {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()},
                    "params": pset, "snplist": [[c.decode(), p] for c, p in snps],
                    "excluded": [[c.decode(), p] for c, p in excluded + extra_excl] if use_excl else [],
                    "sample": sid, "consensus": seq, "line_ends": line_ends,
                })
    finally:
        shutil.rmtree(tmp)
    return runs


def gen_pileup_vectors(captured):
    from snppipeline import pileup
    from oracle import fuzz

    res = doctest.testmod(pileup)
    assert res.failed == 0, res
    vec = {"reference_doctests": {"attempted": res.attempted, "failed": res.failed}}

    # --- strip vectors: doctest examples (pileup.py:294-309), SURVEY A.1 probes, fuzz
    strip_in = [".,.actg,,,", "^K.,.^Fa,,,^K", "$.,.$*$*,,,*", ".,.+10AAAAAAAAAAa,,,", "+2TT.,.+10AAAAAAAAAAa,,,+2GC",
                ".,.-10AAAAAAAAAAa,,,", "-2TT.,.-10AAAAAAAAAAa,,,-2GC", "^Kc-2TT..$a+10AAAAAAAAAAa,,*,-2GC",
                "^^.A", "^+2AC.", "^$.$", "..+9AC", ".-1A+1C.", ".+1A2C", ".+A.", "+1$A", ".+3A-1C.", ".+2A-1CG.",
                "+1+1AA.", ".-2+1A.", "", "^", "+", "5", "+5", "$"]
    rng = random.Random(1234)
    alphabet = "^^^++--$$0123456789.,.,ACGTacgt*"
    for _ in range(6000):
        strip_in.append("".join(rng.choice(alphabet) for _ in range(rng.randint(0, 24))))
    for _ in range(1500):
        strip_in.append(fuzz._adversarial(rng))
    vec["strip"] = [[s, pileup.Record._strip_unwanted_base_patterns(s)] for s in strip_in]

    # --- record vectors: doctest records (pileup.py:98-184, 513-548) + fuzz
    lines = [
        "NC_011149.1\t42\tG\t9\taaAaA+6TAAGAG..+5AAGAG.,\t21G1G-111",
        "ID\t628640\tA\t20\t**.,,.,.............\t22E?;9HF;H8EDGHHI?GH",
        "gi|197247352|ref|NC_011149.1|\t4663812\tT\t0",
        "ID\t1\tA\t20\tTTccAAGG\t22E?;9HF;H8EDGHHI?GH",
        "ID\t1\tA\t20\tTTtccAAAGG\t22E?;9HF;H8EDGHHI?GH",
        "ID\t42\tG\t14\taaaaAAAA...,,,\t00001111222333",
        "ID\t42\tG\t14\taAAAAAAA...,,,\t00001111222333",
        "ID\t42\tG\t14\taaaAAAAA...,,,\t00001111222333",
        "ID\t42\tG\t14\taaaAAA....,,,,\t00011122223333",
        "ID\t42\tg\t14\taaaAAA....,,,,\t00011122223333",
        "ID\t42\tg\t0",
        "ID\t7\tG\t4\t**..\tIIII",
        "ID\t7\tN\t3\t...\tIII",
        "ID\t7\tA\t2\t.\tII",
        "ID\t7\tA\t5\t.....\tIII",
    ]
    rng = random.Random(77)
    for _ in range(4200):
        lines.append(fuzz.fuzz_line(rng))
    vec["records"] = [record_vector(pileup, ln) for ln in lines]

    # --- float-threshold table: cons < depth * freq for the stock frequencies (pileup.py:564)
    thr = {}
    for f in (0.6, 0.75, 0.9, 1.0, 0.51):
        thr[repr(f)] = [min(k for k in range(0, d + 2) if not (k < d * f)) for d in range(0, 400)]
    vec["freq_threshold"] = thr

    vec["runs"] = gen_file_runs(captured, [
        (11, dict(genome_len=3000, n_sites=80), PARAM_SETS[1]),
        (12, dict(genome_len=2500, n_sites=60, contigs=("ctgB", "ctgA", "ctgAA")), PARAM_SETS[2]),
        (13, dict(genome_len=1500, n_sites=40, mean_depth=9), PARAM_SETS[0]),
        (14, dict(genome_len=1200, n_sites=50, mean_depth=70), PARAM_SETS[3]),
    ])
    return vec



def gen_bad_line_runs(captured):
    """Files with ONE odd line through the reference's own call_consensus driver: the exception class it ends with, or the
    consensus when the line is none of its business."""
    from oracle import fuzz
    from snppipeline import call_consensus as cc
    from snppipeline import utils as ref_utils
    runs = []
    tmp = tempfile.mkdtemp(prefix="golden_")
    try:
        for seed, kw in ((31, dict(genome_len=1500, n_sites=50, mean_depth=14)), (32, dict(genome_len=1200, n_sites=40, contigs=("cB", "cA")))):
            base, refs, sites = fuzz.synth_pileup(seed, **kw)
            snps = sorted(sites)
            lpath = os.path.join(tmp, "snplist%d.txt" % seed)
            with open(lpath, "w") as f:
                for c, p in snps:
                    f.write("%s\t%d\t1\tx\n" % (c.decode(), p))
            for scenario in fuzz.BAD_LINE_SCENARIOS:
                data = fuzz.with_bad_line(base, scenario, set(snps))
                sdir = os.path.join(tmp, "s%d_%s" % (seed, scenario))
                os.makedirs(sdir)
                ppath = os.path.join(sdir, "reads.all.pileup")
                with open(ppath, "wb") as f:
                    f.write(data)
                args = argparse.Namespace(
                    snpListFile=lpath, allPileupFile=ppath, consensusFile=os.path.join(sdir, "consensus.fasta"), excludeFile=None,
                    forceFlag=True, vcfFileName=None, vcfRefName="x", vcfAllPos=False, vcfPreserveRefCase=False, vcfFailedSnpGt=".",
                    minBaseQual=0, minConsFreq=0.6, minConsDpth=1, minConsStrdDpth=0, minConsStrdBias=0.0)
                ref_utils.log_verbosity = 0
                sink, old = io.StringIO(), sys.stdout
                sys.stdout = sink
                run = {"seed": seed, "kw": {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}, "scenario": scenario,
                       "snplist": [[c.decode(), p] for c, p in snps], "params": [0, 0.6, 1, 0, 0.0]}
                try:
                    cc.call_consensus(args)
                    run["consensus"] = captured["last"][1]
                except Exception as err:                        # noqa: B902 — the class is the datum
                    run["exception"] = type(err).__name__
                finally:
                    sys.stdout = old
                runs.append(run)
    finally:
        shutil.rmtree(tmp)
    return runs


def gen_filter_runs():
    """The reference's own filter_regions driver (filter_regions.py:17-71, 205-428) on a small cohort: which records of which
    sample it preserves and removes — mode all / each, with and without outgroup samples, two rule sets."""
    from oracle import fuzz
    from snppipeline import filter_regions as fr
    from snppipeline import utils as ref_utils
    runs = []
    for seed in (41, 42):
        lengths, cohort = fuzz.vcf_cohort(seed)
        for mode in ("all", "each"):
            for outgroup in ([], ["smp02", "smp03"]):
                for edge, windows, max_snps in ((500, [1000, 125, 15], [3, 2, 1]), (100, [300], [2])):
                    tmp = tempfile.mkdtemp(prefix="golden_fr_")
                    try:
                        ref = os.path.join(tmp, "ref.fasta")
                        with open(ref, "w") as f:
                            for c, n in lengths.items():
                                f.write(">%s\n%s\n" % (c, "A" * n))
                        dirs = []
                        for name, recs in cohort.items():
                            sd = os.path.join(tmp, name)
                            os.makedirs(sd)
                            with open(os.path.join(sd, "var.flt.vcf"), "w") as f:
                                f.write(fuzz.vcf_text(recs) if recs else "")
                            dirs.append(sd)
                        dirs_file = os.path.join(tmp, "dirs.txt")
                        with open(dirs_file, "w") as f:
                            f.write("\n".join(dirs) + "\n")
                        og = None
                        if outgroup:
                            og = os.path.join(tmp, "outgroup.txt")
                            with open(og, "w") as f:
                                f.write("\n".join(outgroup) + "\n")
                        args = argparse.Namespace(sampleDirsFile=dirs_file, refFastaFile=ref, forceFlag=True, vcfFileName="var.flt.vcf",
                                                  edgeLength=edge, windowSizeList=windows, maxSnpsList=max_snps, outGroupFile=og, mode=mode)
                        ref_utils.log_verbosity = 0
                        os.environ["StopOnSampleError"] = "false"
                        sink, old_out, old_err = io.StringIO(), sys.stdout, sys.stderr
                        sys.stdout = sys.stderr = sink
                        try:
                            fr.filter_regions(args)
                        finally:
                            sys.stdout, sys.stderr = old_out, old_err
                        result = {}
                        for name, recs in cohort.items():
                            out = {}
                            for kind in ("preserved", "removed"):
                                path = os.path.join(tmp, name, "var.flt_%s.vcf" % kind)
                                if not os.path.exists(path):
                                    out[kind] = None
                                    continue
                                rows = [ln.split("\t")[:2] for ln in open(path).read().split("\n") if ln and not ln.startswith("#")]
                                out[kind] = [[c, int(p)] for c, p in rows]
                            result[name] = out
                        runs.append({"seed": seed, "mode": mode, "outgroup": outgroup, "edge": edge, "windows": windows, "max_snps": max_snps,
                                     "result": result})
                    finally:
                        shutil.rmtree(tmp)
    return runs


def gen_merge_runs():
    """The reference's own merge_sites driver (merge_sites.py:12-133) on the cohort of the filter runs: snplist text and the
    filtered list of sample directories, without a limit and with --maxsnps limits that take some samples out (the directory
    of the run is written as $W; the directories are listed in reverse order, as a user's file may be)."""
    from oracle import fuzz
    from snppipeline import merge_sites as ms
    from snppipeline import utils as ref_utils
    runs = []
    for seed in (41, 42):
        lengths, cohort = fuzz.vcf_cohort(seed)
        sizes = sorted(len(set(r)) for r in cohort.values())
        for max_snps in (-1, sizes[len(sizes) // 2], sizes[0], 0):
            tmp = tempfile.mkdtemp(prefix="golden_ms_")
            try:
                dirs = []
                for name, recs in cohort.items():
                    sd = os.path.join(tmp, name)
                    os.makedirs(sd)
                    with open(os.path.join(sd, "var.flt.vcf"), "w") as f:
                        f.write(fuzz.vcf_text(recs + recs[:2]))            # (two records twice: a VCF may repeat a position)
                    dirs.append(sd)
                dirs_file = os.path.join(tmp, "dirs.txt")
                with open(dirs_file, "w") as f:
                    f.write("\n".join(reversed(dirs)) + "\n")
                args = argparse.Namespace(sampleDirsFile=dirs_file, vcfFileName="var.flt.vcf", snpListFile=os.path.join(tmp, "snplist.txt"),
                                          forceFlag=True, maxSnps=max_snps, filteredSampleDirsFile=dirs_file + ".filtered")
                ref_utils.log_verbosity = 0
                sink, old = io.StringIO(), sys.stdout
                sys.stdout = sink
                try:
                    ms.merge_sites(args)
                finally:
                    sys.stdout = old
                runs.append({"seed": seed, "max_snps": max_snps, "snplist": open(args.snpListFile).read(),
                             "filtered": open(args.filteredSampleDirsFile).read().replace(tmp, "$W")})
            finally:
                shutil.rmtree(tmp)
    return runs


def gen_distance_runs():
    """The reference's own distance driver (distance.py:14-118) on untidy SNP matrix files: both TSV texts, or the exception."""
    from oracle import fuzz
    from snppipeline import distance as dm
    from snppipeline import utils as ref_utils
    runs = []
    for name, text in fuzz.untidy_snpmas():
        tmp = tempfile.mkdtemp(prefix="golden_dist_")
        try:
            path = os.path.join(tmp, "snpma.fasta")
            with open(path, "w", newline="") as f:
                f.write(text)
            args = argparse.Namespace(inputFile=path, pairwiseFile=os.path.join(tmp, "p.tsv"), matrixFile=os.path.join(tmp, "m.tsv"), forceFlag=True)
            ref_utils.log_verbosity = 0
            sink, old = io.StringIO(), sys.stdout
            sys.stdout = sink
            run = {"name": name}
            try:
                dm.calculate_snp_distances(args)
                run["pairwise"] = open(args.pairwiseFile).read()
                run["matrix"] = open(args.matrixFile).read()
            except Exception as err:                            # noqa: B902 — the class is the datum
                run["exception"] = type(err).__name__
            finally:
                sys.stdout = old
            runs.append(run)
        finally:
            shutil.rmtree(tmp)
    return runs


LONG_REFS = ["AC", "ac", "Ac", "N,", ".,", ",.", "A.", "gT,", "12", "*A", "a[", "`T", "ACGTNacgtn", ",,", "..", "T,c.G", "zZ", "-+",
             "AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAa"]


def gen_longref_vectors():
    """pileup.Record on lines whose reference-base field has several bytes (pileup.py:223 takes any string; '.' and ','
    are then replaced by the whole upper- / lower-cased field, pileup.py:255-258)."""
    from snppipeline import pileup
    from oracle import fuzz
    rng = random.Random(4242)
    lines = []
    for k in range(600):
        f = fuzz.fuzz_line(rng).split("\t")
        if len(f) < 6:
            continue
        f[2] = LONG_REFS[k % len(LONG_REFS)]
        lines.append("\t".join(f))
    lines += ["ID\t42\tGC\t14\taaaAAA....,,,,\t00011122223333", "ID\t43\tg,\t6\t..,,AC\tIIIIII", "ID\t44\tAC\t0", "ID\t45\tAC\t3\t.,.\t!!!"]
    # depth columns that int() takes and the 32 unsigned bits of the device record do not hold (pileup.py:225): the reference
    # compares the value with 0 and prints it
    wide = []
    for k, depth in enumerate(["-3", "-0", "+7", "4294967295", "4294967296", "5000000000", "-5000000000", "1_000", "-1_0", "00012",
                               "4611686018427387903", "-4611686018427387903"]):
        wide.append("DP\t%d\t%s\t%s\t..,,AaCc*\tIIII5IIII" % (100 + k, "ACGTN"[k % 5], depth))
        wide.append("DP\t%d\tAC\t%s\t..,,Gg\tIIIIII" % (200 + k, depth))
        wide.append("DP\t%d\tT\t%s" % (300 + k, depth))
    # fields longer than the 64 bytes a spill record of the device holds (they go on in the records behind it: 1 640 bytes each):
    # 65, 66, the last byte of one more record, the first of two more, thousands; with '.', ',' and both cases inside
    very = []
    for k, n in enumerate([65, 66, 100, 129, 500, 64 + 1640, 64 + 1640 + 1, 64 + 2 * 1640, 5000, 20000]):
        # (',' inside the field makes the reference's replace — and the oracle's — quadratic: only in the shorter ones)
        body = "".join(rng.choice("ACGTNacgtn" + (".," if k % 2 and n < 2000 else "")) for _ in range(n))
        very.append("LR\t%d\t%s\t12\t..,,AaCc.,*G\tIIII5III!III" % (400 + k, body))
        very.append("LR\t%d\t%s\t0" % (500 + k, body))
        f = fuzz.fuzz_line(rng).split("\t")
        while len(f) < 6:
            f = fuzz.fuzz_line(rng).split("\t")
        f[0], f[1], f[2] = "LR", str(600 + k), body
        very.append("\t".join(f))
    return {"records": [record_vector(pileup, ln) for ln in lines], "wide_depth_records": [record_vector(pileup, ln) for ln in wide],
            "very_long_ref_records": [record_vector(pileup, ln) for ln in very]}


def gen_steps_vectors():
    from snppipeline import filter_regions as fr
    from snppipeline import utils as ru

    assert doctest.testmod(fr).failed == 0
    vec = {}
    rng = random.Random(5)
    dense = []
    fixed = [(3, 1000, []), (3, 1000, [1, 2, 3, 1001]), (3, 1000, [1, 20, 30, 1000]), (3, 1000, [1, 20, 30, 40, 1000]),
             (3, 1000, [1, 20, 30, 40, 501, 600, 1000, 1500]), (3, 1000, [1, 2, 3, 1000, 1500, 3001, 3002, 3003, 4000])]
    for m, w, s in fixed:
        dense.append({"m": m, "w": w, "snps": s, "out": [list(t) for t in fr.find_dense_regions(m, w, s)]})
    for _ in range(400):
        n = rng.randint(0, 60)
        span = rng.choice([200, 2000, 20000])
        s = sorted(rng.randint(1, span) for _ in range(n))       # duplicates allowed
        m, w = rng.choice([(3, 1000), (2, 125), (1, 15), (5, 50), (1, 1)])
        dense.append({"m": m, "w": w, "snps": s, "out": [list(t) for t in fr.find_dense_regions(m, w, s)]})
    vec["find_dense_regions"] = dense

    merges = []
    for _ in range(400):
        regs = []
        for _ in range(rng.randint(0, 14)):
            a = rng.randint(0, 300)
            regs.append((a, a + rng.randint(0, 40)))
        merges.append({"in": [list(r) for r in regs], "out": [list(r) for r in ru.merge_regions(list(regs))]})
    vec["merge_regions"] = merges

    inreg = []
    for _ in range(300):
        regs = ru.merge_regions([(a, a + rng.randint(0, 10)) for a in (rng.randint(0, 100) for _ in range(rng.randint(0, 6)))])
        p = rng.randint(0, 115)
        inreg.append({"pos": p, "regions": [list(r) for r in regs], "out": bool(ru.in_region(p, regs))})
    vec["in_region"] = inreg

    dist = []
    alpha = "ACGTacgt-NnRY*"
    for _ in range(500):
        n = rng.randint(0, 90)
        a = "".join(rng.choice(alpha) for _ in range(n))
        b = "".join(rng.choice(alpha) for _ in range(n))
        dist.append({"a": a, "b": b, "d": ru.calculate_sequence_distance(a, b)})
    vec["sequence_distance"] = dist

    # collect_dense_regions with a stand-in reader (objects with CHROM/POS)
    class R(object):
        def __init__(self, c, p):
            self.CHROM, self.POS = c, p
    coll = []
    for _ in range(120):
        lens = {"c1": rng.choice([900, 1000, 1001, 5000, 50000]), "c2": 30000}
        samples = []
        for s in range(rng.randint(1, 4)):
            recs = []
            for _ in range(rng.randint(0, 40)):
                c = rng.choice(["c1", "c1", "c2", "c3"])
                recs.append([c, rng.randint(1, lens.get(c, 60000))])
            samples.append(recs)
        edge = rng.choice([500, 100, 1])
        rules = rng.choice([([3, 2, 1], [1000, 125, 15]), ([3], [1000]), ([1], [5])])
        bad = {}
        for recs in samples:
            fr.collect_dense_regions([R(c, p) for c, p in recs], bad, lens, edge, rules[0], rules[1])
        merged = {c: [list(r) for r in ru.merge_regions(v)] for c, v in bad.items()}
        coll.append({"lens": lens, "samples": samples, "edge": edge, "max_snps": rules[0], "windows": rules[1], "out": merged})
    vec["collect_all"] = coll

    # snplist writer
    d = {("chrB", 5): ["s1"], ("chrA", 100): ["s2", "s1"], ("chrA", 20): ["s3"], ("chrAA", 3): ["s1", "s2", "s3"]}
    tmp = tempfile.mkdtemp()
    try:
        p = os.path.join(tmp, "snplist.txt")
        ru.write_list_of_snps(p, d)
        with open(p) as f:
            vec["snplist_writer"] = {"in": [[k[0], k[1], v] for k, v in d.items()], "out": f.read()}
        vec["snplist_reader"] = [list(t) for t in ru.read_snp_position_list(p)]
    finally:
        shutil.rmtree(tmp)
    return vec


def gen_metrics_vectors():
    """The depth-column sum of collect_metrics.py:325-340 (avePileupDepth), by running the reference's own
    collect_metrics() on sample directories that hold nothing but a pileup; with a 1-base reference the printed
    ``%.2f`` average is the sum itself.  Also missingPos (collect_metrics.py:109-128 counts the '-' of a FASTA record)."""
    install_stubs()
    from snppipeline import collect_metrics as cm
    from oracle import fuzz
    out = []
    specs = [("synth", dict(seed=5, genome_len=5000, n_sites=20)), ("synth", dict(seed=6, genome_len=3000, n_sites=10, mean_depth=120)),
             ("synth", dict(seed=7, genome_len=2500, n_sites=10, contigs=("NODE_2", "NODE_10"))),
             ("odd", dict(seed=9, eol="\n")), ("odd", dict(seed=9, eol="\r\n")), ("odd", dict(seed=10, eol="\n"))]
    old = os.environ.get("StopOnSampleError")
    os.environ["StopOnSampleError"] = "false"
    try:
        for kind, kw in specs:
            if kind == "synth":
                k2 = dict(kw)
                if "contigs" in k2:
                    k2["contigs"] = tuple(k2["contigs"])
                data = fuzz.synth_pileup(**k2)[0]
            else:
                data = fuzz.odd_depth_lines(kw["seed"], kw["eol"].encode())
            tmp = tempfile.mkdtemp()
            try:
                sd = os.path.join(tmp, "samples", "s1")
                os.makedirs(sd)
                with open(os.path.join(sd, "reads.all.pileup"), "wb") as f:
                    f.write(data)
                ref = os.path.join(tmp, "ref.fasta")
                with open(ref, "w") as f:
                    f.write(">r\nA\n")
                args = argparse.Namespace(referenceFile=ref, sampleDir=sd, consensusFastaFileName="consensus.fasta",
                                          consensusPreservedFastaFileName="consensus_preserved.fasta", consensusVcfFileName="consensus.vcf",
                                          consensusPreservedVcfFileName="consensus_preserved.vcf", maxSnps=-1,
                                          metricsFile=os.path.join(sd, "metrics"), forceFlag=True, verbose=0, subparser_name="collect_metrics")
                stdout = sys.stdout
                sys.stdout = io.StringIO()
                try:
                    cm.collect_metrics(args)
                finally:
                    sys.stdout = stdout
                props = dict(line.rstrip("\n").split("=", 1) for line in open(os.path.join(sd, "metrics")) if "=" in line)
                out.append({"kind": kind, "kw": kw, "avePileupDepth": props["avePileupDepth"], "bytes": len(data)})
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
    finally:
        if old is None:
            os.environ.pop("StopOnSampleError", None)
        else:
            os.environ["StopOnSampleError"] = old
    return {"depth_sum": out}


def copy_fixtures():
    """Data files from the reference's bundled ExpectedResults trees."""
    src = os.path.join(REF, "snppipeline", "data")
    dst = os.path.join(GOLD, "fixtures")
    if os.path.isdir(dst):
        shutil.rmtree(dst)
    keep_top = ("snplist.txt", "snplist_preserved.txt", "snpma.fasta", "snpma_preserved.fasta",
                "snp_distance_matrix.tsv", "snp_distance_matrix_preserved.tsv",
                "snp_distance_pairwise.tsv", "snp_distance_pairwise_preserved.tsv",
                "referenceSNP.fasta", "referenceSNP_preserved.fasta", "metrics.tsv")
    keep_sample = ("var.flt.vcf", "var.flt_preserved.vcf", "var.flt_removed.vcf", "consensus.fasta",
                   "consensus_preserved.fasta", "consensus.vcf", "consensus_preserved.vcf", "metrics")
    for ds in ("lambdaVirus", "agona", "listeria"):
        exp = os.path.join(src, ds + "ExpectedResults")
        out = os.path.join(dst, ds)
        os.makedirs(out)
        buf = io.BytesIO()
        with tarfile.open(fileobj=buf, mode="w") as tar:
            for name in keep_top:
                p = os.path.join(exp, name)
                if os.path.isfile(p):
                    tar.add(p, arcname=name)
            sdir = os.path.join(exp, "samples")
            for s in sorted(os.listdir(sdir)):
                for name in keep_sample:
                    p = os.path.join(sdir, s, name)
                    if os.path.isfile(p):
                        tar.add(p, arcname="samples/%s/%s" % (s, name))
        with open(os.path.join(out, "expected.tar.xz"), "wb") as f:
            import lzma
            f.write(lzma.compress(buf.getvalue(), preset=9))
        # contig ids + lengths of the reference FASTA (filter_regions only needs these)
        inp = os.path.join(src, ds + "Inputs", "reference")
        lens = {}
        if os.path.isdir(inp):
            for fn in os.listdir(inp):
                name, n = None, 0
                with open(os.path.join(inp, fn)) as f:
                    for line in f:
                        if line.startswith(">"):
                            if name is not None:
                                lens[name] = n
                            name, n = line[1:].split()[0], 0
                        else:
                            n += len(line.strip())
                if name is not None:
                    lens[name] = n
        sl = os.path.join(src, ds + "Inputs", "sampleList")
        meta = {"contig_lengths": lens}
        if os.path.isfile(sl):
            meta["sampleList"] = open(sl).read().split()
        with open(os.path.join(out, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
    shutil.copy(os.path.join(src, "lambdaVirusInputs", "reference", "lambda_virus.fasta"), os.path.join(dst, "lambdaVirus"))
    # the listeria reference (3 MB of FASTA text) pins snp_reference on a second data set: kept xz-compressed
    import lzma
    with open(os.path.join(src, "listeriaInputs", "reference", "CFSAN023463.HGAP.draft.fasta"), "rb") as f:
        raw = f.read()
    with open(os.path.join(dst, "listeria", "CFSAN023463.HGAP.draft.fasta.xz"), "wb") as f:
        f.write(lzma.compress(raw, preset=9))


CLI_LINES = [
    "filter_regions dirs.txt ref.fasta",
    "filter_regions -f -n var.flt.vcf dirs.txt ref.fasta --edge_length 500 --window_size 1000 125 15 --max_snp 3 2 1 --mode all",
    "filter_regions dirs.txt ref.fasta -l 10 -w 100 -m 2 -g outgroup.txt -M each -v 0",
    "merge_sites dirs.txt dirs.filtered",
    "merge_sites -f -n var.flt_preserved.vcf -o snplist_preserved.txt --maxsnps 1000 dirs.txt dirs.filtered",
    "call_consensus reads.all.pileup",
    "call_consensus -f -l snplist.txt -o s/consensus.fasta --vcfRefName ref.fasta --minConsFreq 0.6 --minConsDpth 3 --vcfFileName consensus.vcf s/reads.all.pileup",
    "call_consensus -l snplist_preserved.txt -o s/consensus_preserved.fasta -e s/var.flt_removed.vcf -q 15 -c 0.9 -D 5 -d 2 -b 0.1 --vcfFailedSnpGt 1 --vcfPreserveRefCase s/reads.all.pileup",
    "snp_matrix dirs.txt",
    "snp_matrix -f -c consensus_preserved.fasta -o snpma_preserved.fasta dirs.txt",
    "distance snpma.fasta",
    "distance -f -p pairs.tsv -m matrix.tsv snpma.fasta",
    "snp_reference reference/lambda_virus.fasta",
    "snp_reference -f -v 0 -l snplist_preserved.txt -o referenceSNP_preserved.fasta reference/lambda_virus.fasta",
]


def gen_cli_vectors():
    """argparse results of the reference's own parser for the subcommands this build provides."""
    import types as _t
    jr = _t.ModuleType("jobrunner")                      # orchestration only (run.py); never called here
    jr.JobRunner = object
    jr.JobRunnerException = Exception
    sys.modules.setdefault("jobrunner", jr)
    from snppipeline import cfsan_snp_pipeline as ref_cli
    out = []
    for line in CLI_LINES:
        ns = vars(ref_cli.parse_command_line(line))
        clean = {k: v for k, v in ns.items() if k not in ("func", "excepthook")}
        clean["excepthook"] = ns["excepthook"].__name__ if ns.get("excepthook") else None
        clean["func"] = ns["func"].__name__
        out.append({"line": line, "args": clean})
    return out


def dump(name, obj):
    raw = json.dumps(obj, separators=(",", ":"), sort_keys=True).encode()
    with open(os.path.join(GOLD, name), "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0) as g:
            g.write(raw)
    print(name, len(raw), "bytes raw")


def _runs3(captured):
    # line ends: the reference reads the pileup in text mode (universal newlines: "\n", "\r\n" and a lone "\r" end a line;
    # '\v' / '\f' do not, they are whitespace to str.split()) — files of a few dozen scan tiles through its own driver
    runs = []
    for variant in ("crlf", "mixed", "vt_ff", "repeats"):
        runs += gen_file_runs(captured, [(21, dict(genome_len=2600, n_sites=90, mean_depth=18), PARAM_SETS[1]),
                                         (22, dict(genome_len=1800, n_sites=60, contigs=("NODE_1_length_419034_cov_23.1", "c")), PARAM_SETS[2])],
                              line_ends=variant)
    return {"runs": runs}


def _utf8names(captured):
    # contig names that are not plain ASCII (the reference reads the pileup as text: they are just names to it), one with a '~'
    runs = gen_file_runs(captured, [(31, dict(genome_len=1500, n_sites=50, contigs=("chr\u00e4", "\u67d3\u8272\u4f531", "a~b")), PARAM_SETS[1]),
                                    (32, dict(genome_len=2200, n_sites=70, mean_depth=14, contigs=("\u00e9coli_K12", "z")), PARAM_SETS[2])])
    return {"runs": runs}


def _runs2(captured):
    # later additions: shapes the device kernels treat specially (512-byte lane window, long contig names,
    # positions around the powers of ten), again through the reference's own driver
    long_names = ("NODE_1_length_419034_cov_23.1", "scaffold_with_a_very_long_name_that_exceeds_44_bytes_000001", "c")
    runs = gen_file_runs(captured, [
        (15, dict(genome_len=900, n_sites=60, mean_depth=140), PARAM_SETS[0]),
        (16, dict(genome_len=1500, n_sites=70, contigs=long_names), PARAM_SETS[1]),
        (17, dict(genome_len=10400, n_sites=150), PARAM_SETS[2]),
        (18, dict(genome_len=800, n_sites=50, mean_depth=220), PARAM_SETS[4]),
    ], extra_sites=[p10 + d for p10 in (10, 100, 1000, 10000) for d in (-1, 0, 1)])
    return {"runs": runs}


# every committed vector file: slice name -> (file under tests/golden/, generator taking the captured-writes dict).
# tests/test_golden_regen.py regenerates each one into a scratch directory and compares it with the committed file.
SLICES = {
    "pileup": ("pileup_vectors.json.gz", gen_pileup_vectors),
    "steps": ("steps_vectors.json.gz", lambda captured: gen_steps_vectors()),
    "cli": ("cli_vectors.json.gz", lambda captured: gen_cli_vectors()),
    "metrics": ("metrics_vectors.json.gz", lambda captured: gen_metrics_vectors()),
    "longref": ("longref_vectors.json.gz", lambda captured: gen_longref_vectors()),
    "runs2": ("pileup_runs2.json.gz", _runs2),
    "runs3": ("pileup_runs3.json.gz", _runs3),
    "utf8names": ("pileup_runs_utf8.json.gz", _utf8names),
    "distance": ("distance_runs.json.gz", lambda captured: {"runs": gen_distance_runs()}),
    "merge": ("merge_runs.json.gz", lambda captured: {"runs": gen_merge_runs()}),
    "filter": ("filter_runs.json.gz", lambda captured: {"runs": gen_filter_runs()}),
    "badlines": ("badline_runs.json.gz", lambda captured: {"runs": gen_bad_line_runs(captured)}),
}


def main():
    """``gen_golden.py`` writes every slice and the fixture trees; ``--only NAME`` one slice (or ``fixtures``); ``--out DIR``
    writes there instead of tests/golden/ (what the regeneration test does)."""
    global GOLD
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=sorted(SLICES) + ["fixtures"])
    ap.add_argument("--out")
    opt = ap.parse_args()
    if opt.out:
        GOLD = os.path.abspath(opt.out)
    os.makedirs(GOLD, exist_ok=True)
    captured = install_stubs()
    if opt.only == "fixtures":
        copy_fixtures()
        return
    for name in ([opt.only] if opt.only else list(SLICES)):
        fname, gen = SLICES[name]
        dump(fname, gen(captured))
    if not opt.only:
        copy_fixtures()


if __name__ == "__main__":
    main()

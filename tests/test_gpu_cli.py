"""The five hot subcommands end to end (CLI -> host mirror -> C ABI -> HIP kernels) against the reference's bundled
ExpectedResults trees and, for call_consensus (no pileup ships with the reference), against the oracle."""
import filecmp
import os
import shutil
import time

import pytest

from oracle import fuzz
from oracle import pileup_oracle as po
from oracle import vcf_oracle as vo

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _argv(monkeypatch):
    monkeypatch.setattr("sys.argv", ["cfsan_snp_pipeline", "test"])


def _run(line):
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    args = cli.parse_command_line(line)
    args.verbose = 0
    assert cli.run_command_from_args(args) == 0


def _work_tree(tmp_path, fixture_root, meta, dataset):
    """Copy the fixture samples into a scratch work dir; returns (work dir, sample dirs file, reference fasta)."""
    work = str(tmp_path / "work")
    shutil.copytree(os.path.join(fixture_root, "samples"), os.path.join(work, "samples"))
    dirs = sorted(os.path.join(work, "samples", s) for s in os.listdir(os.path.join(work, "samples")))
    dirs_file = os.path.join(work, "sampleDirectories.txt")
    with open(dirs_file, "w") as f:
        f.write("\n".join(reversed(dirs)) + "\n")                 # deliberately unsorted
    ref = os.path.join(work, "reference.fasta")
    if dataset == "lambdaVirus":
        from tests.conftest import GOLD
        shutil.copy(os.path.join(GOLD, "fixtures", "lambdaVirus", "lambda_virus.fasta"), ref)
    else:
        with open(ref, "w") as f:
            lens = meta["contig_lengths"] or {"contig_missing_from_this_checkout": 100}
            for name, n in lens.items():
                f.write(">%s\n" % name)
                for i in range(0, n, 60000):
                    f.write("N" * min(60000, n - i) + "\n")
    return work, dirs_file, ref


@pytest.mark.parametrize("ds", ["lambdaVirus", "agona", "listeria"])
def test_filter_merge_matrix_distance_on_bundled_trees(tmp_path, fixture_trees, ds):
    root, meta = fixture_trees[ds]
    work, dirs_file, ref = _work_tree(tmp_path, root, meta, ds)
    samples = sorted(os.listdir(os.path.join(work, "samples")))
    # remove the expected outputs that the commands below must recreate
    for s in samples:
        for name in ("var.flt_preserved.vcf", "var.flt_removed.vcf"):
            p = os.path.join(work, "samples", s, name)
            if os.path.exists(p):
                os.remove(p)
    # step 5: filter_regions with the pipeline's default parameters (snppipeline.conf:211)
    _run("filter_regions -n var.flt.vcf %s %s --edge_length 500 --window_size 1000 125 15 --max_snp 3 2 1 --mode all" % (dirs_file, ref))
    for s in samples:
        for name in ("var.flt_preserved.vcf", "var.flt_removed.vcf"):
            want = os.path.join(root, "samples", s, name)
            if os.path.exists(want):
                assert filecmp.cmp(os.path.join(work, "samples", s, name), want, shallow=False), (s, name)
    if ds == "lambdaVirus":
        # freshness (filter_regions.py:246-256): nothing is rewritten while every target is newer than every source; one newer
        # source VCF makes all targets stale in mode all
        probe = [os.path.join(work, "samples", s, name) for s in samples for name in ("var.flt_preserved.vcf", "var.flt_removed.vcf")]
        before = [os.stat(p).st_mtime_ns for p in probe]
        _run("filter_regions -n var.flt.vcf %s %s --edge_length 500 --window_size 1000 125 15 --max_snp 3 2 1 --mode all" % (dirs_file, ref))
        assert [os.stat(p).st_mtime_ns for p in probe] == before
        future = time.time() + 5
        os.utime(os.path.join(work, "samples", samples[-1], "var.flt.vcf"), (future, future))
        _run("filter_regions -n var.flt.vcf %s %s --edge_length 500 --window_size 1000 125 15 --max_snp 3 2 1 --mode all" % (dirs_file, ref))
        assert all(a != b for a, b in zip([os.stat(p).st_mtime_ns for p in probe], before))
        past = time.time() - 60
        os.utime(os.path.join(work, "samples", samples[-1], "var.flt.vcf"), (past, past))
        for s in samples:
            for name in ("var.flt_preserved.vcf", "var.flt_removed.vcf"):
                assert filecmp.cmp(os.path.join(work, "samples", s, name), os.path.join(root, "samples", s, name), shallow=False)
    # steps 6.1 / 6.2: merge_sites
    _run("merge_sites -n var.flt.vcf -o %s/snplist.txt %s %s.OrigVCF.filtered" % (work, dirs_file, dirs_file))
    _run("merge_sites -n var.flt_preserved.vcf -o %s/snplist_preserved.txt %s %s.PresVCF.filtered" % (work, dirs_file, dirs_file))
    assert filecmp.cmp(work + "/snplist.txt", root + "/snplist.txt", shallow=False)
    assert filecmp.cmp(work + "/snplist_preserved.txt", root + "/snplist_preserved.txt", shallow=False)
    assert open(dirs_file + ".OrigVCF.filtered").read() == open(dirs_file).read()
    # steps 8.x: snp_matrix from the bundled consensus files, steps 11.x: distance
    for suffix in ("", "_preserved"):
        have_cons = all(os.path.exists(os.path.join(work, "samples", s, "consensus%s.fasta" % suffix)) for s in samples)
        snpma = "%s/snpma%s.fasta" % (work, suffix)
        if have_cons:
            _run("snp_matrix -c consensus%s.fasta -o %s %s.OrigVCF.filtered" % (suffix, snpma, dirs_file))
            assert filecmp.cmp(snpma, "%s/snpma%s.fasta" % (root, suffix), shallow=False)
        else:
            shutil.copy("%s/snpma%s.fasta" % (root, suffix), snpma)
        _run("distance -p %s/pairs%s.tsv -m %s/matrix%s.tsv %s" % (work, suffix, work, suffix, snpma))
        assert filecmp.cmp("%s/matrix%s.tsv" % (work, suffix), "%s/snp_distance_matrix%s.tsv" % (root, suffix), shallow=False)
        pw = "%s/snp_distance_pairwise%s.tsv" % (root, suffix)
        if os.path.exists(pw):
            assert filecmp.cmp("%s/pairs%s.tsv" % (work, suffix), pw, shallow=False)
    # freshness: a second run without -f leaves the outputs alone
    before = os.stat(work + "/snplist.txt").st_mtime_ns
    _run("merge_sites -n var.flt.vcf -o %s/snplist.txt %s %s.OrigVCF.filtered" % (work, dirs_file, dirs_file))
    assert os.stat(work + "/snplist.txt").st_mtime_ns == before


def test_filter_regions_mode_each_and_outgroup(tmp_path, fixture_trees):
    from oracle import steps_oracle as so
    from snp_pipeline_amd import utils
    root, meta = fixture_trees["lambdaVirus"]
    work, dirs_file, ref = _work_tree(tmp_path, root, meta, "lambdaVirus")
    og = os.path.join(work, "outgroup.txt")
    with open(og, "w") as f:
        f.write("sample2\n")
    _run("filter_regions -f %s %s -l 500 -w 1000 125 15 -m 3 2 1 -M each -g %s" % (dirs_file, ref, og))
    lens = meta["contig_lengths"]
    for s in sorted(os.listdir(os.path.join(work, "samples"))):
        d = os.path.join(work, "samples", s)
        header, data, sites = utils.read_vcf_sites(os.path.join(d, "var.flt.vcf"))
        pres = utils.read_vcf_sites(os.path.join(d, "var.flt_preserved.vcf"))[2]
        rem = utils.read_vcf_sites(os.path.join(d, "var.flt_removed.vcf"))[2]
        if s == "sample2":
            assert filecmp.cmp(os.path.join(d, "var.flt.vcf"), os.path.join(d, "var.flt_preserved.vcf"), shallow=False) and rem == []
            continue
        bad = so.bad_regions([(s, sites)], lens, 500, [3, 2, 1], [1000, 125, 15], mode="each")[s]
        assert rem == [k for k in sites if so.in_region(k[1], bad[k[0]])]
        assert pres == [k for k in sites if not so.in_region(k[1], bad[k[0]])]


def test_call_consensus_cli_vs_oracle(tmp_path):
    """call_consensus subcommand on synthetic pileups: consensus.fasta and consensus.vcf against the oracle."""
    work = tmp_path
    for seed, kw, extra in ((41, dict(genome_len=5000, n_sites=120), ""),
                            (42, dict(genome_len=2500, n_sites=70, contigs=("ctgB", "ctgA")), " -q 15 -c 0.9 -D 5 -d 2 -b 0.1 --vcfFailedSnpGt 1")):
        data, _, sites = fuzz.synth_pileup(seed, **kw)
        sdir = work / ("sample%d" % seed)
        sdir.mkdir()
        (sdir / "reads.all.pileup").write_bytes(data)
        snps = sorted(sites + [(sites[0][0], 77_000_000)])
        excl = sites[::6] + [(sites[0][0], 9)]
        with open(str(work / ("snplist%d.txt" % seed)), "w") as f:
            for c, p in snps:
                f.write("%s\t%d\t1\tsampleX\n" % (c.decode(), p))
        with open(str(sdir / "var.flt_removed.vcf"), "w") as f:
            f.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n")
            for c, p in excl:
                f.write("%s\t%d\t.\tA\tC\t.\tPASS\t.\tGT\t1/1\n" % (c.decode(), p))
        params = po.CallerParams(15, 0.9, 5, 2, 0.1) if extra else po.CallerParams(0, 0.6, 3, 0, 0.0)
        base_flags = extra if extra else " --minConsFreq 0.6 --minConsDpth 3"
        _run("call_consensus -l %s/snplist%d.txt -o %s/consensus.fasta -e %s/var.flt_removed.vcf --vcfRefName ref.fasta "
             "--vcfFileName consensus.vcf%s %s/reads.all.pileup" % (work, seed, sdir, sdir, base_flags, sdir))
        want, detail = po.call_consensus_sites(data, snps, set(excl), params)
        want_fa = ">sample%d\n" % seed + "".join(want.decode()[i:i + 60] + "\n" for i in range(0, len(want), 60))
        assert (sdir / "consensus.fasta").read_text() == want_fa
        names = po.filter_names(params)
        rows = []
        for _, line in po.iter_lines(data):                            # pileup order, one row per parsed position
            f = line.split()
            key = (f[0], int(f[1]))
            if key in detail:
                rec, base, mask = detail[key]
                failed = [names[i] for i in range(6) if mask >> i & 1] or None
                rows.append(vo.vcf_row(rec, failed, "1" if extra else "."))
        got = [ln for ln in (sdir / "consensus.vcf").read_text().split("\n") if ln and not ln.startswith("#")]
        assert got == rows
    # empty snplist: exit 0, FASTA with only the header line (regression_tests.sh:3156-3207)
    (work / "empty.txt").write_text("")
    _run("call_consensus -f -l %s/empty.txt -o %s/empty.fasta %s/sample41/reads.all.pileup" % (work, work, work))
    assert (work / "empty.fasta").read_text() == ">sample41\n"


def test_call_consensus_cli_error_paths(tmp_path, monkeypatch):
    """Missing snplist -> exit 100 always; missing pileup -> 100 / 98 by StopOnSampleError; corrupt snplist -> exception hook."""
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    log = tmp_path / "error.log"
    monkeypatch.setenv("errorOutputFile", str(log))
    pile = tmp_path / "s1" / "reads.all.pileup"
    pile.parent.mkdir()
    pile.write_bytes(b"c\t1\tA\t1\t.\tI\n")
    monkeypatch.setenv("StopOnSampleError", "false")
    with pytest.raises(SystemExit) as ei:
        _run("call_consensus -l %s/absent.txt -o %s/c.fasta %s" % (tmp_path, tmp_path, pile))
    assert ei.value.code == 100 and "cannot call consensus without the snplist file" in log.read_text()
    (tmp_path / "snplist.txt").write_text("c\t1\t1\ts1\n")
    with pytest.raises(SystemExit) as ei:
        _run("call_consensus -l %s/snplist.txt -o %s/c.fasta %s/absent.pileup" % (tmp_path, tmp_path, tmp_path))
    assert ei.value.code == 98
    monkeypatch.delenv("StopOnSampleError")
    with pytest.raises(SystemExit) as ei:
        _run("call_consensus -l %s/snplist.txt -o %s/c.fasta %s/absent.pileup" % (tmp_path, tmp_path, tmp_path))
    assert ei.value.code == 100
    (tmp_path / "corrupt.txt").write_text("c\tnot_a_number\n")
    with pytest.raises(ValueError):
        _run("call_consensus -f -l %s/corrupt.txt -o %s/c.fasta %s" % (tmp_path, tmp_path, pile))
    # a pileup that is not valid UTF-8 ends the reference's text-mode read with UnicodeDecodeError: same exception class here;
    # valid multi-byte characters are refused as what they are
    from snp_pipeline_amd.device import PileupFormatError
    pile.write_bytes(b"c\t1\tA\t3\t.\xff.\tIII\n")
    with pytest.raises(UnicodeDecodeError):
        _run("call_consensus -f -l %s/snplist.txt -o %s/c.fasta %s" % (tmp_path, tmp_path, pile))
    pile.write_bytes("c\t1\tA\t3\t.\u00e9.\tIII\n".encode("utf-8"))
    with pytest.raises(PileupFormatError):
        _run("call_consensus -f -l %s/snplist.txt -o %s/c.fasta %s" % (tmp_path, tmp_path, pile))


def test_console_script_subprocess_without_torch(tmp_path):
    """bin/cfsan_snp_pipeline as run.py starts it: a separate process, which loads the HIP library without importing
    torch (seconds saved per sample process) and writes the same FASTA."""
    import subprocess
    import sys
    import time
    data, _, sites = fuzz.synth_pileup(43, genome_len=4000, n_sites=90)
    sdir = tmp_path / "sampleS"
    sdir.mkdir()
    (sdir / "reads.all.pileup").write_bytes(data)
    with open(str(tmp_path / "snplist.txt"), "w") as f:
        for c, p in sites:
            f.write("%s\t%d\t1\tsampleS\n" % (c.decode(), p))
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "cfsan_snp_pipeline")
    probe = ("import runpy, sys\n"
             "sys.argv = ['cfsan_snp_pipeline', 'call_consensus', '-v', '0', '-l', %r, '-o', %r, '--minConsDpth', '3', %r]\n"
             "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert not e.code, e.code\n"
             "print('torch' in sys.modules)"
             % (str(tmp_path / "snplist.txt"), str(sdir / "consensus.fasta"), str(sdir / "reads.all.pileup"), exe))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().endswith("False"), r.stdout[-500:]            # torch was never imported
    want, _ = po.call_consensus_sites(data, sites, set(), po.CallerParams(0, 0.6, 3, 0, 0.0))
    assert (sdir / "consensus.fasta").read_text() == ">sampleS\n" + "".join(want.decode()[i:i + 60] + "\n" for i in range(0, len(want), 60))
    print("console script wall time %.2f s" % (time.time() - t0))


def test_call_consensus_records_metrics_byproducts(tmp_path):
    """--amdMetricsRefFasta: avePileupDepth (depth-column sum of the scan / reference length, "%.2f", collect_metrics.py:325-340)
    and missingPos / missingPosPreserved go to the sample's metrics file, other lines kept."""
    data, ref, sites = fuzz.synth_pileup(47, genome_len=3000, n_sites=80)
    sdir = tmp_path / "sampleM"
    sdir.mkdir()
    (sdir / "reads.all.pileup").write_bytes(data)
    fa = tmp_path / "ref.fasta"
    fa.write_text("".join(">%s some description\n%s\n" % (c.decode() if isinstance(c, bytes) else c, "\n".join(seq[i:i + 70] for i in range(0, len(seq), 70)))
                          for c, seq in ref.items()))
    ref_len = sum(len(seq) for seq in ref.values())
    with open(str(tmp_path / "snplist.txt"), "w") as f:
        for c, p in sites:
            f.write("%s\t%d\t1\tsampleM\n" % (c.decode(), p))
    (sdir / "metrics").write_text('sample="sampleM"\nmissingPos=999\n')
    _run("call_consensus -v 0 -l %s/snplist.txt -o %s/consensus.fasta --minConsDpth 3 --amdMetricsRefFasta %s %s/reads.all.pileup" % (tmp_path, sdir, fa, sdir))
    want, _ = po.call_consensus_sites(data, sites, set(), po.CallerParams(0, 0.6, 3, 0, 0.0))
    props = dict(ln.split("=", 1) for ln in (sdir / "metrics").read_text().split("\n") if "=" in ln)
    assert props == {"sample": '"sampleM"', "missingPos": str(want.count(b"-")), "avePileupDepth": "%.2f" % (po.depth_sum(data) / float(ref_len))}
    # the preserved flow records its own key, in the file named by --amdMetricsFile
    with open(str(sdir / "excl.vcf"), "w") as f:
        f.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n")
        for c, p in sites[::4]:
            f.write("%s\t%d\t.\tA\tC\t.\tPASS\t.\tGT\t1/1\n" % (c.decode(), p))
    _run("call_consensus -v 0 -l %s/snplist.txt -o %s/consensus_preserved.fasta -e %s/excl.vcf --minConsDpth 3 --amdMetricsRefFasta %s "
         "--amdMetricsFile %s/m2 %s/reads.all.pileup" % (tmp_path, sdir, sdir, fa, tmp_path, sdir))
    want2, _ = po.call_consensus_sites(data, sites, set(sites[::4]), po.CallerParams(0, 0.6, 3, 0, 0.0))
    props2 = dict(ln.split("=", 1) for ln in (tmp_path / "m2").read_text().split("\n") if "=" in ln)
    assert props2["missingPosPreserved"] == str(want2.count(b"-")) and "missingPos" not in props2


def test_consensus_vcf_rows_for_repeated_positions(tmp_path):
    """A pileup that lists a position twice: the FASTA takes the last line (call_consensus.py:171-176), consensus.vcf gets a
    row for every matching line in file order (:178-180)."""
    lines = [b"c1\t5\tA\t3\tGGG\tIII", b"c1\t6\tC\t3\t...\tIII", b"c1\t5\tA\t4\tTTTt\tIIII", b"c1\t7\tG\t2\t..\tII", b"c1\t9\tG\t3\taaa\tIII",
             b"c1\t9\tG\t1\t.\tI"]
    data = b"\n".join(lines) + b"\n"
    sdir = tmp_path / "dup"
    sdir.mkdir()
    (sdir / "reads.all.pileup").write_bytes(data)
    (tmp_path / "snplist.txt").write_text("c1\t5\t1\tdup\nc1\t7\t1\tdup\nc1\t9\t1\tdup\n")
    _run("call_consensus -v 0 -l %s/snplist.txt -o %s/consensus.fasta --vcfFileName consensus.vcf %s/reads.all.pileup" % (tmp_path, sdir, sdir))
    assert (sdir / "consensus.fasta").read_text() == ">dup\nTGG\n"
    params = po.CallerParams(0, 0.6, 1, 0, 0.0)
    names = po.filter_names(params)
    want = []
    for ln in (lines[0], lines[2], lines[3], lines[4], lines[5]):
        rec = po.parse_record(po.split_fields(ln), 0)
        base, mask = po.call_record(rec, params)
        want.append(vo.vcf_row(rec, [names[i] for i in range(6) if mask >> i & 1] or None, "."))
    got = [x for x in (sdir / "consensus.vcf").read_text().split("\n") if x and not x.startswith("#")]
    assert got == want


def test_positions_with_more_than_eight_symbols_get_all_their_alt_alleles(tmp_path):
    """pileup.Record ranks any number of distinct symbols (pileup.py:259-266) and consensus.vcf lists every one that is not the
    reference as an ALT allele with its depths (vcf_writer.py:317-331).  The per-site record keeps eight; the rest comes back
    through the context's spill.  Read bases with IUPAC codes give 10 to 16 symbols here, and every fifth line has a reference
    field of two or three bytes (REF shows the string, pileup.py:223, vcf_writer.py:295): rows of the per-sample command, of
    --vcfAllPos and of the batch command against the oracle's writer."""
    import random
    rng = random.Random(5)
    alphabet = "ACGTNRYKMSWBDHV*"
    lines, keys = [], []
    for i in range(40):
        pos = 100 + 7 * i
        k = rng.choice((3, 6, 9, 10, 12, 16))
        syms = rng.sample(alphabet, k)
        reads = [rng.choice(syms) for _ in range(rng.randint(k, 60))] + syms
        rng.shuffle(reads)
        bases = "".join(c.lower() if (c != "*" and rng.random() < 0.5) else c for c in reads)
        quals = "".join(chr(33 + rng.randint(0, 40)) for _ in reads)
        if i % 5 == 4:                                              # every '.' / ',' spelled out by a reference field of several bytes
            bases = bases.replace("*", ".").replace("n", ",")
        ref = rng.choice(("AC", "g,", "Tn", "ac.")) if i % 5 == 4 else rng.choice("ACGTacgt")
        # (round 4) a depth column int() takes and 32 unsigned bits do not hold: the reference only prints it (pileup.py:225)
        depth = "-%d" % len(reads) if i % 7 == 3 else ("%d" % (5_000_000_000 + i) if i % 7 == 5 else "%d" % len(reads))
        lines.append("ctg1\t%d\t%s\t%s\t%s\t%s" % (pos, ref, depth, bases, quals))
        keys.append((b"ctg1", pos))
    data = ("\n".join(lines) + "\n").encode()
    params = po.CallerParams(10, 0.6, 3, 0, 0.0)
    want, detail = po.call_consensus_sites(data, keys, set(), params)
    assert sum(1 for rec, _, _ in detail.values() if len(rec.most_common_good_bases or []) > 8) >= 10
    names = po.filter_names(params)
    rows = []
    for key in keys:
        rec, base, mask = detail[key]
        rows.append(vo.vcf_row(rec, [names[i] for i in range(6) if mask >> i & 1] or None, "."))
    assert any(row.split("\t")[4].count(",") >= 9 for row in rows)             # ten or more ALT alleles in a row
    assert sum(1 for row in rows if len(row.split("\t")[3]) > 1) == 8          # REF strings
    assert sum(1 for row in rows if row.split("\t")[9].split(":")[1].startswith("-")) == 6 and sum(1 for row in rows if row.split("\t")[9].split(":")[1].startswith("50000000")) == 5
    with open(str(tmp_path / "snplist.txt"), "w") as f:
        for c, p in keys:
            f.write("%s\t%d\t1\ts\n" % (c.decode(), p))
    flags = "--minBaseQual 10 --minConsFreq 0.6 --minConsDpth 3 --vcfRefName ref.fasta --vcfFileName consensus.vcf"
    dirs = []
    for name in ("sA", "sB"):
        sdir = tmp_path / name
        sdir.mkdir()
        (sdir / "reads.all.pileup").write_bytes(data)
        dirs.append(str(sdir))
    fa = ">sA\n" + "".join(want.decode()[i:i + 60] + "\n" for i in range(0, len(want), 60))

    def data_rows(path):
        return [ln for ln in open(path).read().split("\n") if ln and not ln.startswith("#")]

    _run("call_consensus -l %s/snplist.txt -o %s/consensus.fasta %s %s/reads.all.pileup" % (tmp_path, dirs[0], flags, dirs[0]))
    assert open(dirs[0] + "/consensus.fasta").read() == fa and data_rows(dirs[0] + "/consensus.vcf") == rows
    _run("call_consensus -f --vcfAllPos -l %s/snplist.txt -o %s/consensus.fasta %s %s/reads.all.pileup" % (tmp_path, dirs[0], flags, dirs[0]))
    assert data_rows(dirs[0] + "/consensus.vcf") == rows
    (tmp_path / "dirs.txt").write_text("\n".join(dirs) + "\n")
    for d in dirs:
        for n in ("consensus.fasta", "consensus.vcf"):
            if os.path.exists(os.path.join(d, n)):
                os.remove(os.path.join(d, n))
    _run("call_consensus_batch -l %s/snplist.txt %s %s/dirs.txt" % (tmp_path, flags, tmp_path))
    for d in dirs:
        assert data_rows(d + "/consensus.vcf") == rows


def test_filter_regions_runs_of_the_reference_driver(tmp_path):
    """filter_runs.json.gz through the console script: the records each sample keeps and loses, as the reference's own driver
    decided them (mode all / each, outgroup samples, two rule sets)."""
    from tests.conftest import load_golden
    for k, run in enumerate(load_golden("filter_runs.json.gz")["runs"]):
        lengths, cohort = fuzz.vcf_cohort(run["seed"])
        work = tmp_path / ("run%d" % k)
        work.mkdir()
        ref = work / "ref.fasta"
        ref.write_text("".join(">%s\n%s\n" % (c, "A" * n) for c, n in lengths.items()))
        dirs = []
        for name, recs in cohort.items():
            sd = work / name
            sd.mkdir()
            (sd / "var.flt.vcf").write_text(fuzz.vcf_text(recs))
            dirs.append(str(sd))
        (work / "dirs.txt").write_text("\n".join(dirs) + "\n")
        extra = ""
        if run["outgroup"]:
            (work / "outgroup.txt").write_text("\n".join(run["outgroup"]) + "\n")
            extra = " -g %s/outgroup.txt" % work
        _run("filter_regions -f -n var.flt.vcf --edge_length %d --window_size %s --max_snp %s --mode %s%s %s/dirs.txt %s"
             % (run["edge"], " ".join(map(str, run["windows"])), " ".join(map(str, run["max_snps"])), run["mode"], extra, work, ref))
        for name in cohort:
            for kind in ("preserved", "removed"):
                rows = [ln.split("\t")[:2] for ln in (work / name / ("var.flt_%s.vcf" % kind)).read_text().split("\n") if ln and not ln.startswith("#")]
                assert [[c, int(p)] for c, p in rows] == run["result"][name][kind], (k, run["mode"], run["outgroup"], name, kind)


def test_merge_sites_runs_of_the_reference_driver(tmp_path):
    """merge_runs.json.gz through the console script: snplist.txt and the filtered list of sample directories byte for byte
    as the reference's own driver wrote them (no limit, --maxsnps limits that take some / all but one / all samples out)."""
    from tests.conftest import load_golden
    for k, run in enumerate(load_golden("merge_runs.json.gz")["runs"]):
        _, cohort = fuzz.vcf_cohort(run["seed"])
        work = tmp_path / ("run%d" % k)
        work.mkdir()
        dirs = []
        for name, recs in cohort.items():
            sd = work / name
            sd.mkdir()
            (sd / "var.flt.vcf").write_text(fuzz.vcf_text(recs + recs[:2]))
            dirs.append(str(sd))
        (work / "dirs.txt").write_text("\n".join(reversed(dirs)) + "\n")
        _run("merge_sites -f -n var.flt.vcf --maxsnps %d -o %s/snplist.txt %s/dirs.txt %s/dirs.txt.filtered" % (run["max_snps"], work, work, work))
        assert (work / "snplist.txt").read_text() == run["snplist"], (run["seed"], run["max_snps"])
        assert (work / "dirs.txt.filtered").read_text().replace(str(work), "$W") == run["filtered"], (run["seed"], run["max_snps"])


def test_more_spill_positions_than_the_first_arena_holds(tmp_path):
    """Round 3 refused a call in which more than 1 024 positions needed a spill record (here: every listed position has a
    reference-base field of two bytes, 3 000 of them, some with a depth outside 32 bits).  The context's arena now grows and the
    call is repeated: rows of the per-sample command, of --vcfAllPos and of the batch command against the oracle's writer."""
    import random
    rng = random.Random(11)
    lines, keys = [], []
    for i in range(3000):
        pos = 10 + 3 * i
        n = rng.randint(3, 30)
        bases = "".join(rng.choice(".,.,AaCcGgTt*") for _ in range(n))
        quals = "".join(chr(33 + rng.randint(5, 40)) for _ in range(n))
        depth = "-%d" % n if i % 97 == 0 else "%d" % n
        ref = rng.choice(("AC", "gT", "N,", "ac."))
        if i % 50 == 7:                             # fields longer than a spill record's 64 bytes: they take the records behind it too
            ref = "".join(rng.choice("ACGTacgtN.,") for _ in range(rng.choice((65, 200, 1704, 1705, 4000))))
        lines.append("ctgX\t%d\t%s\t%s\t%s\t%s" % (pos, ref, depth, bases, quals))
        keys.append((b"ctgX", pos))
    data = ("\n".join(lines) + "\n").encode()
    params = po.CallerParams(10, 0.6, 3, 0, 0.0)
    want, detail = po.call_consensus_sites(data, keys, set(), params)
    names = po.filter_names(params)
    rows = []
    for key in keys:
        rec, base, mask = detail[key]
        rows.append(vo.vcf_row(rec, [names[i] for i in range(6) if mask >> i & 1] or None, "."))
    with open(str(tmp_path / "snplist.txt"), "w") as f:
        for c, p in keys:
            f.write("%s\t%d\t1\ts\n" % (c.decode(), p))
    flags = "--minBaseQual 10 --minConsFreq 0.6 --minConsDpth 3 --vcfRefName ref.fasta --vcfFileName consensus.vcf"
    dirs = []
    for name in ("sA", "sB"):
        sdir = tmp_path / name
        sdir.mkdir()
        (sdir / "reads.all.pileup").write_bytes(data)
        dirs.append(str(sdir))

    def data_rows(path):
        return [ln for ln in open(path).read().split("\n") if ln and not ln.startswith("#")]

    _run("call_consensus -l %s/snplist.txt -o %s/consensus.fasta %s %s/reads.all.pileup" % (tmp_path, dirs[0], flags, dirs[0]))
    assert data_rows(dirs[0] + "/consensus.vcf") == rows
    assert open(dirs[0] + "/consensus.fasta").read() == ">sA\n" + "".join(want.decode()[i:i + 60] + "\n" for i in range(0, len(want), 60))
    _run("call_consensus -f --vcfAllPos -l %s/snplist.txt -o %s/consensus.fasta %s %s/reads.all.pileup" % (tmp_path, dirs[0], flags, dirs[0]))
    assert data_rows(dirs[0] + "/consensus.vcf") == rows
    (tmp_path / "dirs.txt").write_text("\n".join(dirs) + "\n")
    _run("call_consensus_batch -f -l %s/snplist.txt %s %s/dirs.txt" % (tmp_path, flags, tmp_path))
    for d in dirs:
        assert data_rows(d + "/consensus.vcf") == rows


def test_contig_names_that_are_not_ascii(tmp_path, capfd):
    """A pileup whose contig names are UTF-8 text ("chrä", "染色体1") is just a pileup to the reference, which reads it as text
    (golden runs through its own driver: pileup_runs_utf8.json.gz).  Round 3 refused it; now the per-sample command hands the
    device a copy with the names escaped to ASCII (utf8_names.py) and spells the CHROM column of consensus.vcf back.  Still
    refused, loudly: non-ASCII characters in any other column."""
    from snp_pipeline_amd.device import PileupFormatError
    from tests.conftest import load_golden
    for n, run in enumerate(load_golden("pileup_runs_utf8.json.gz")["runs"]):
        kw = dict(run["kw"])
        kw["contigs"] = tuple(kw["contigs"])
        data, _, _ = fuzz.synth_pileup(run["seed"], **kw)
        sdir = tmp_path / ("s%d" % n)
        sdir.mkdir()
        (sdir / "reads.all.pileup").write_bytes(data)
        snps = [(c.encode(), p) for c, p in run["snplist"]]
        excl = [(c.encode(), p) for c, p in run["excluded"]]
        with open(str(sdir / "snplist.txt"), "w", encoding="utf-8") as f:
            for c, p in run["snplist"]:
                f.write("%s\t%d\t1\tx\n" % (c, p))
        with open(str(sdir / "excl.vcf"), "w", encoding="utf-8") as f:
            f.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n")
            for c, p in run["excluded"]:
                f.write("%s\t%d\t.\tA\tC\t.\tPASS\t.\tGT\t1/1\n" % (c, p))
        q, c_, D, d_, b = run["params"]
        flags = "-q %d -c %g -D %d -d %d -b %g --vcfRefName ref.fasta --vcfFileName consensus.vcf" % (q, c_, D, d_, b)
        more = " -e %s/excl.vcf" % sdir if excl else ""
        params = po.CallerParams(*run["params"])
        want, detail = po.call_consensus_sites(data, snps, set(excl), params)
        assert want.decode() == run["consensus"]
        names = po.filter_names(params)
        rows = []
        for _, line in po.iter_lines(data):
            f = line.split()
            key = (f[0], int(f[1]))
            if key in detail:
                rec, base, mask = detail[key]
                rows.append(vo.vcf_row(rec, [names[i] for i in range(6) if mask >> i & 1] or None, "."))
        for all_pos in ("", " --vcfAllPos"):
            _run("call_consensus -f -l %s/snplist.txt -o %s/consensus.fasta%s %s%s %s/reads.all.pileup" % (sdir, sdir, more, flags, all_pos, sdir))
            fa = (sdir / "consensus.fasta").read_text()
            assert fa == ">s%d\n" % n + "".join(run["consensus"][i:i + 60] + "\n" for i in range(0, len(run["consensus"]), 60)), (n, all_pos)
            got = [ln for ln in (sdir / "consensus.vcf").read_text(encoding="utf-8").split("\n") if ln and not ln.startswith("#")]
            if not all_pos:
                assert got == rows and any(r.startswith("chrä\t") or r.startswith("écoli_K12\t") for r in got), (n, all_pos)
            else:                                                   # a row for every line of the pileup, names spelled back
                assert len(got) == len(list(po.iter_lines(data))) and set(r.split("\t")[0] for r in got) == set(kw["contigs"])
        assert [p for p in os.listdir(str(sdir)) if "snpgpu_names" in p] == []
        # the batch form: this sample beside a plain-ASCII one, same snplist; the one with the names takes the bridge alone
        one = {}
        for all_pos in ("", " --vcfAllPos"):
            _run("call_consensus -f -l %s/snplist.txt -o %s/consensus.fasta%s %s%s %s/reads.all.pileup" % (sdir, sdir, more, flags, all_pos, sdir))
            one[all_pos] = ((sdir / "consensus.fasta").read_bytes(), (sdir / "consensus.vcf").read_bytes())
        plain = tmp_path / ("p%d" % n)
        plain.mkdir()
        pdata, _, _ = fuzz.synth_pileup(run["seed"] + 1, **dict(kw, contigs=("plainA", "plainB")))
        (plain / "reads.all.pileup").write_bytes(pdata)
        (plain / "excl.vcf").write_bytes((sdir / "excl.vcf").read_bytes())
        pwant, _ = po.call_consensus_sites(pdata, snps, set(excl), params)
        (tmp_path / "dirs.txt").write_text("%s\n%s\n" % (plain, sdir))
        for all_pos in ("", " --vcfAllPos"):
            os.unlink(str(sdir / "consensus.fasta"))
            _run("call_consensus_batch -f -l %s/snplist.txt%s %s%s %s/dirs.txt" % (sdir, " -e excl.vcf" if excl else "", flags, all_pos, tmp_path))
            assert ((sdir / "consensus.fasta").read_bytes(), (sdir / "consensus.vcf").read_bytes()) == one[all_pos], (n, all_pos)
            assert (plain / "consensus.fasta").read_text() == ">p%d\n" % n + "".join(
                pwant.decode()[i:i + 60] + "\n" for i in range(0, len(pwant), 60))
        assert [p for p in os.listdir(str(sdir)) if "snpgpu_names" in p] == []
    # a non-ASCII character where text and bytes part ways: refused, not guessed
    sdir = tmp_path / "bad"
    sdir.mkdir()
    (sdir / "reads.all.pileup").write_bytes("c\t5\tA\t3\t.ä.\tIII\n".encode("utf-8"))
    (sdir / "snplist.txt").write_text("c\t5\t1\tx\n")
    with pytest.raises(PileupFormatError) as ei:
        _run("call_consensus -l %s/snplist.txt -o %s/consensus.fasta %s/reads.all.pileup" % (sdir, sdir, sdir))
    assert "outside the contig-name column" in str(ei.value)
    capfd.readouterr()

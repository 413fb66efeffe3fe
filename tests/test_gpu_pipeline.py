"""The post-alignment pipeline end to end through the console script, the way run.py chains it (run.py:672-784, steps 4-11),
on a small synthetic outbreak: call_sites -> filter_regions -> merge_sites -> call_consensus -> snp_matrix -> distance ->
snp_reference, each stage reading the files the stage before wrote.  Every artifact is compared with the chain of CPU
restatements fed with the same pileups (varscan_oracle -> steps_oracle -> pileup_oracle -> steps_oracle)."""
import os
import time

import pytest

from oracle import fuzz
from oracle import pileup_oracle as po
from oracle import steps_oracle as so
from oracle import varscan_oracle as vo

pytestmark = pytest.mark.gpu


def _run(line):
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    args = cli.parse_argument_list([w.replace("\x00", " ") for w in line.split()])        # (\x00: a blank inside one argument)
    args.verbose = 0
    assert cli.run_command_from_args(args) == 0


def _fasta(name, seq):
    return ">%s\n" % name + "".join(seq[i:i + 60] + "\n" for i in range(0, len(seq), 60))


def test_pipeline_stages_chained_on_a_synthetic_outbreak(tmp_path, monkeypatch):
    refs, piles = fuzz.cohort_pileups(7, n_samples=6, genome_len=12000, mean_depth=22, n_scattered=9)
    names = ["iso%02d" % i for i in range(len(piles))]
    work = tmp_path
    ref_path = work / "reference" / "ref.fasta"
    ref_path.parent.mkdir()
    ref_path.write_text("".join(_fasta(c, refs[c]) for c in refs))
    old = time.time() - 1000
    os.utime(str(ref_path), (old, old))
    dirs = []
    for name, data in zip(names, piles):
        sdir = work / "samples" / name
        sdir.mkdir(parents=True)
        bam = sdir / "reads.sorted.deduped.indelrealigned.bam"
        bam.write_bytes(b"placeholder: the pileup below is newer, so samtools is not run (call_sites.py:70-72)")
        os.utime(str(bam), (old, old))
        (sdir / "reads.all.pileup").write_bytes(data)
        dirs.append(str(sdir))
    dirs_file = str(work / "sampleDirectories.txt")
    with open(dirs_file, "w") as f:
        f.write("\n".join(reversed(dirs)) + "\n")
    monkeypatch.setenv("VarscanMpileup2snp_ExtraParams", "--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5")
    monkeypatch.chdir(work)

    # ---- step 4 (second half): call_sites -> var.flt.vcf ----
    vprm = vo.Params(**vo.PIPELINE_DEFAULTS)
    sites = {}
    for name, data, sdir in zip(names, piles, dirs):
        _run("call_sites %s %s" % (ref_path, sdir))
        want = vo.mpileup2snp(data, vprm)
        assert open(os.path.join(sdir, "var.flt.vcf")).read() == want, name
        sites[name] = [(ln.split("\t")[0], int(ln.split("\t")[1])) for ln in want.splitlines() if not ln.startswith("#")]
        assert len(sites[name]) > 5
    # ---- step 5: filter_regions ----
    lens = {c: len(refs[c]) for c in refs}
    _run("filter_regions -n var.flt.vcf %s %s --edge_length 100 --window_size 1000 125 15 --max_snp 3 2 1 --mode all" % (dirs_file, ref_path))
    bad = so.bad_regions([(n, sites[n]) for n in names], lens, 100, [3, 2, 1], [1000, 125, 15], mode="all")
    kept, removed = {}, {}
    for name, sdir in zip(names, dirs):
        kept[name] = [k for k in sites[name] if not so.in_region(k[1], bad.get(k[0], []))]
        removed[name] = [k for k in sites[name] if so.in_region(k[1], bad.get(k[0], []))]
        src = [ln for ln in open(os.path.join(sdir, "var.flt.vcf")).read().splitlines(True) if not ln.startswith("#")]
        for fname, keys in (("var.flt_preserved.vcf", kept[name]), ("var.flt_removed.vcf", removed[name])):
            got = [ln for ln in open(os.path.join(sdir, fname)).read().splitlines(True) if not ln.startswith("#")]
            keyset = set(keys)
            assert got == [ln for ln in src if (ln.split("\t")[0], int(ln.split("\t")[1])) in keyset], (name, fname)
    assert sum(len(v) for v in removed.values()) > 10 and sum(len(v) for v in kept.values()) > 25
    # ---- step 6: merge_sites on the preserved files ----
    snplist = str(work / "snplist_preserved.txt")
    _run("merge_sites -n var.flt_preserved.vcf -o %s %s %s.filtered" % (snplist, dirs_file, dirs_file))
    merged, excluded = so.merge_sites([(d, n, kept[n]) for d, n in sorted(zip(dirs, names))])
    assert open(snplist).read() == so.snplist_text(merged) and not excluded
    snp_keys = [(k[0].encode(), k[1]) for k, _ in merged]
    # ---- step 7: call_consensus per sample, with the removed sites as the exclude file, and a consensus.vcf ----
    cprm = po.CallerParams(15, 0.9, 5, 2, 0.1)
    seqs = {}
    for name, data, sdir in zip(names, piles, dirs):
        _run("call_consensus -l %s -e %s/var.flt_removed.vcf -o %s/consensus_preserved.fasta -q 15 -c 0.9 -D 5 -d 2 -b 0.1 "
             "--vcfRefName ref.fasta --vcfFileName consensus_preserved.vcf %s/reads.all.pileup" % (snplist, sdir, sdir, sdir))
        want, _ = po.call_consensus_sites(data, snp_keys, set((c.encode(), p) for c, p in removed[name]), cprm)
        seqs[name] = want.decode()
        assert open(os.path.join(sdir, "consensus_preserved.fasta")).read() == _fasta(name, seqs[name]), name
        rows = [ln for ln in open(os.path.join(sdir, "consensus_preserved.vcf")).read().splitlines() if not ln.startswith("#")]
        # one row per parsed position that has a pileup line: the snplist and this sample's excluded sites (call_consensus.py:150-180)
        got_keys = [(r.split("\t")[0].encode(), int(r.split("\t")[1])) for r in rows]
        parsed = set(snp_keys) | set((c.encode(), p) for c, p in removed[name])
        present = set((ln.split(b"\t")[0], int(ln.split(b"\t")[1])) for ln in data.split(b"\n") if ln)
        assert got_keys == sorted(parsed & present), name
        assert all(("Region" in r.split(":")[-1]) == (k in set((c.encode(), p) for c, p in removed[name])) for r, k in zip(rows, got_keys))
    # ---- steps 8 / 11: snp_matrix, distance ----
    snpma = str(work / "snpma_preserved.fasta")
    _run("snp_matrix -c consensus_preserved.fasta -o %s %s.filtered" % (snpma, dirs_file))
    assert open(snpma).read() == "".join(_fasta(n, seqs[n]) for n in sorted(names))
    _run("distance -p %s/pairs.tsv -m %s/matrix.tsv %s" % (work, work, snpma))
    ids, table = so.distance_tables(seqs)
    assert open(str(work / "pairs.tsv")).read() == so.pairwise_text(ids, table)
    assert open(str(work / "matrix.tsv")).read() == so.matrix_text(ids, table)
    assert max(table.values()) > 10                         # the two clades are apart
    # ---- step 9: snp_reference ----
    _run("snp_reference -l %s -o %s/referenceSNP_preserved.fasta %s" % (snplist, work, ref_path))
    want_ref = ""
    for c in sorted(refs):
        bases = "".join(refs[c][p - 1].upper() for k, p in [kk for kk, _ in merged] if k == c)
        want_ref += _fasta(c, bases)                         # (a record for every contig, with or without positions: utils.py:1103-1110)
    assert open(str(work / "referenceSNP_preserved.fasta")).read() == want_ref


# ---------------------------------------------------------------------------------------------------------------------------
# hot_path_batch: the same chain as ONE job (every pileup over the host link once) must write the same bytes as the separate
# subcommands run.py starts (run.py:662-784): call_sites, filter_regions, merge_sites x 2, call_consensus x 2 per sample,
# snp_matrix x 2, snp_reference x 2, distance x 2.
# ---------------------------------------------------------------------------------------------------------------------------
PER_SAMPLE = ("var.flt.vcf", "var.flt_preserved.vcf", "var.flt_removed.vcf", "consensus.fasta", "consensus.vcf",
              "consensus_preserved.fasta", "consensus_preserved.vcf")
TOP_LEVEL = ("snplist.txt", "snplist_preserved.txt", "sampleDirectories.txt.OrigVCF.filtered", "sampleDirectories.txt.PresVCF.filtered",
             "snpma.fasta", "snpma_preserved.fasta", "snp_distance_pairwise.tsv", "snp_distance_matrix.tsv",
             "snp_distance_pairwise_preserved.tsv", "snp_distance_matrix_preserved.tsv", "referenceSNP.fasta", "referenceSNP_preserved.fasta")
VARSCAN_EXTRA = "--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5"
CONSENSUS_EXTRA = "-q 15 -c 0.9 -D 5 -d 2 -b 0.1"


def _outbreak_tree(work, seed=7, n_samples=6, genome_len=12000, contigs=("ctg1", "ctg2")):
    refs, piles = fuzz.cohort_pileups(seed, n_samples=n_samples, genome_len=genome_len, mean_depth=22, n_scattered=9, contigs=contigs)
    names = ["iso%02d" % i for i in range(len(piles))]
    ref_path = work / "reference" / "ref.fasta"
    ref_path.parent.mkdir()
    ref_path.write_text("".join(_fasta(c, refs[c]) for c in refs))
    old = time.time() - 1000
    os.utime(str(ref_path), (old, old))
    dirs = []
    for name, data in zip(names, piles):
        sdir = work / "samples" / name
        sdir.mkdir(parents=True)
        bam = sdir / "reads.sorted.deduped.indelrealigned.bam"
        bam.write_bytes(b"placeholder: the pileup below is newer, so samtools is not run (call_sites.py:70-72)")
        os.utime(str(bam), (old, old))
        (sdir / "reads.all.pileup").write_bytes(data)
        dirs.append(str(sdir))
    dirs_file = str(work / "sampleDirectories.txt")
    with open(dirs_file, "w") as f:
        f.write("\n".join(reversed(dirs)) + "\n")
    return str(ref_path), dirs, dirs_file, piles


def test_hot_path_batch_without_consensus_vcf_writes_the_other_files_alike(tmp_path, monkeypatch):
    """--noConsensusVcf: no per-site count records are made (the lane kernels without them, the group check on the filter bytes
    instead of the records' status bytes) — every other file as the full job writes it, and a malformed line at a listed
    position is the same sample error."""
    work = tmp_path
    ref_path, dirs, dirs_file, piles = _outbreak_tree(work, seed=13, n_samples=5, genome_len=6000)
    monkeypatch.setenv("VarscanMpileup2snp_ExtraParams", VARSCAN_EXTRA)
    monkeypatch.setenv("StopOnSampleError", "false")
    monkeypatch.setenv("errorOutputFile", str(work / "error.log"))
    monkeypatch.chdir(work)
    filter_extra = "--edge_length 100 --window_size 1000 125 15 --max_snp 3 2 1 --mode all"
    line = ("hot_path_batch -f %s %s --filterRegionsExtraParams=%s --callConsensusExtraParams=%s"
            % (dirs_file, ref_path, filter_extra.replace(" ", "\x00"), CONSENSUS_EXTRA.replace(" ", "\x00")))
    _run(line)
    want = _snapshot(work, dirs)
    _run(line + " --noConsensusVcf")
    for sdir in dirs:
        for name in PER_SAMPLE:
            path = os.path.join(sdir, name)
            if name.endswith("consensus.vcf") or name.endswith("consensus_preserved.vcf"):
                assert not os.path.exists(path), path
            else:
                assert open(path, "rb").read() == want[os.path.join(os.path.basename(sdir), name)], path
    for name in TOP_LEVEL:
        assert open(os.path.join(str(work), name), "rb").read() == want[name], name
    # a line no Record can be built from, at a listed position of one sample: that sample fails either way, the others go on
    snplist = [ln.split("\t") for ln in want["snplist.txt"].decode().splitlines()]
    chrom, pos = snplist[len(snplist) // 2][0], snplist[len(snplist) // 2][1]
    p = os.path.join(dirs[1], "reads.all.pileup")
    stamp = os.stat(p)
    data = open(p, "rb").read().split(b"\n")
    for i, ln in enumerate(data):
        f = ln.split(b"\t")
        if len(f) > 3 and f[0] == chrom.encode() and f[1] == pos.encode():
            data[i] = b"\t".join(f[:3] + [b"notanumber"] + f[4:])
            break
    else:
        raise AssertionError("the listed position has no line in sample 1")
    open(p, "wb").write(b"\n".join(data))
    os.utime(p, ns=(stamp.st_atime_ns, stamp.st_mtime_ns))     # (its var.flt.vcf stays the newer file: --siteCalling existing takes it)
    logs = []
    for extra in ("", " --noConsensusVcf"):
        if os.path.exists(str(work / "error.log")):
            os.remove(str(work / "error.log"))
        _run(line + extra + " --siteCalling existing")
        logs.append(open(str(work / "error.log")).read())
        assert "call_consensus failed for sample %s" % os.path.basename(dirs[1]) in logs[-1]
    assert logs[0] == logs[1]


def test_hot_path_batch_with_contig_names_that_are_not_ascii(tmp_path, monkeypatch):
    """Contig names are just names to the reference (it reads its files as text): `chr\u00e4`, `\u67d3\u8272\u4f531`, a name with a `~` in it.
    The one job must write what the separate subcommands write (which take such names sample by sample since round 4) — refused up
    to round 4, now carried the same way: the device sees every name escaped, what is written keeps the names as they are.  Half
    of the samples resident, half streamed."""
    work = tmp_path
    ref_path, dirs, dirs_file, piles = _outbreak_tree(work, seed=11, contigs=("a~b", "chr\u00e4", "\u67d3\u8272\u4f531"), genome_len=7000)
    monkeypatch.setenv("VarscanMpileup2snp_ExtraParams", VARSCAN_EXTRA)
    monkeypatch.chdir(work)
    filter_extra = "--edge_length 100 --window_size 1000 125 15 --max_snp 3 2 1 --mode all"
    _separate_steps(work, ref_path, dirs, dirs_file, filter_extra, "")
    want = _snapshot(work, dirs)
    lists = want["snplist.txt"].decode("utf-8"), want["snplist_preserved.txt"].decode("utf-8")
    assert "chr\u00e4" in lists[0] and "\u67d3\u8272\u4f531" in lists[1] and "a~b" in lists[0]
    for resident in (0, int(2.5 * max(len(p) for p in piles))):
        _run("hot_path_batch -f %s %s --filterRegionsExtraParams=%s --callConsensusExtraParams=%s%s"
             % (dirs_file, ref_path, filter_extra.replace(" ", "\x00"), CONSENSUS_EXTRA.replace(" ", "\x00"),
                " --residentBytes %d" % resident if resident else ""))
        _compare(_snapshot(work, dirs, remove=False), want)
    # the batch form of step 4 alone (its device pass works on bytes: a name is whatever stands in front of the first TAB)
    for sdir in dirs:
        os.remove(os.path.join(sdir, "var.flt.vcf"))
    _run("call_sites_batch %s %s" % (ref_path, dirs_file))
    for sdir in dirs:
        assert open(os.path.join(sdir, "var.flt.vcf"), "rb").read() == want[os.path.join(os.path.basename(sdir), "var.flt.vcf")], sdir


def _separate_steps(work, ref_path, dirs, dirs_file, filter_extra, merge_extra, consensus_extra=None):
    """What run.py:662-784 runs, one subcommand after the other."""
    consensus_extra = consensus_extra or CONSENSUS_EXTRA
    for sdir in dirs:
        _run("call_sites %s %s" % (ref_path, sdir))
    _run("filter_regions -f -n var.flt.vcf %s %s %s" % (dirs_file, ref_path, filter_extra))
    _run("merge_sites -f -n var.flt.vcf -o %s/snplist.txt %s %s %s.OrigVCF.filtered" % (work, merge_extra, dirs_file, dirs_file))
    _run("merge_sites -f -n var.flt_preserved.vcf -o %s/snplist_preserved.txt %s %s %s.PresVCF.filtered" % (work, merge_extra, dirs_file, dirs_file))
    for sdir in dirs:
        _run("call_consensus -f -l %s/snplist.txt -o %s/consensus.fasta --vcfRefName ref.fasta %s --vcfFileName consensus.vcf %s/reads.all.pileup"
             % (work, sdir, consensus_extra, sdir))
        _run("call_consensus -f -l %s/snplist_preserved.txt -o %s/consensus_preserved.fasta -e %s/var.flt_removed.vcf --vcfRefName ref.fasta %s "
             "--vcfFileName consensus_preserved.vcf %s/reads.all.pileup" % (work, sdir, sdir, consensus_extra, sdir))
    for suffix, flt in (("", "OrigVCF"), ("_preserved", "PresVCF")):
        _run("snp_matrix -f -c consensus%s.fasta -o %s/snpma%s.fasta %s.%s.filtered" % (suffix, work, suffix, dirs_file, flt))
        _run("snp_reference -f -l %s/snplist%s.txt -o %s/referenceSNP%s.fasta %s" % (work, suffix, work, suffix, ref_path))
        _run("distance -f -p %s/snp_distance_pairwise%s.tsv -m %s/snp_distance_matrix%s.tsv %s/snpma%s.fasta" % (work, suffix, work, suffix, work, suffix))


def _snapshot(work, dirs, remove=True):
    out = {}
    for sdir in dirs:
        for name in PER_SAMPLE:
            path = os.path.join(sdir, name)
            out[os.path.join(os.path.basename(sdir), name)] = open(path, "rb").read()
            if remove:
                os.remove(path)
    for name in TOP_LEVEL:
        path = os.path.join(str(work), name)
        out[name] = open(path, "rb").read()
        if remove:
            os.remove(path)
    return out


def _compare(got, want):
    assert sorted(got) == sorted(want)
    for name in sorted(want):
        assert got[name] == want[name], name


@pytest.mark.parametrize("filter_extra, maxsnps, partial", [
    ("--edge_length 100 --window_size 1000 125 15 --max_snp 3 2 1 --mode all", False, False),
    ("--edge_length 100 --window_size 1000 125 15 --max_snp 3 2 1 --mode each", True, False),
    ("--edge_length 100 --window_size 1000 125 15 --max_snp 3 2 1 --mode all", False, True),    # two of the six pileups fit: the rest is streamed twice
])
def test_hot_path_batch_equals_the_separate_steps(tmp_path, monkeypatch, filter_extra, maxsnps, partial):
    from snp_pipeline_amd import hot_path
    work = tmp_path
    ref_path, dirs, dirs_file, piles = _outbreak_tree(work)
    monkeypatch.setenv("VarscanMpileup2snp_ExtraParams", VARSCAN_EXTRA)
    monkeypatch.chdir(work)
    merge_extra = ""
    if maxsnps:                                                # a threshold that takes some samples out of the lists, not all
        counts = []
        for sdir in dirs:
            _run("call_sites %s %s" % (ref_path, sdir))
            counts.append(sum(1 for ln in open(os.path.join(sdir, "var.flt.vcf")) if not ln.startswith("#")))
        merge_extra = "--maxsnps %d" % sorted(counts)[len(counts) // 2]
    _separate_steps(work, ref_path, dirs, dirs_file, filter_extra, merge_extra)
    want = _snapshot(work, dirs)
    assert len(want["snplist.txt"].splitlines()) > len(want["snplist_preserved.txt"].splitlines()) > 10
    if maxsnps:
        assert 0 < len(want["sampleDirectories.txt.OrigVCF.filtered"].splitlines()) < len(dirs)
    resident = int(2.5 * max(len(p) for p in piles)) if partial else 0
    _run("hot_path_batch -f %s %s --filterRegionsExtraParams=%s --mergeSitesExtraParams=%s --callConsensusExtraParams=%s%s"
         % (dirs_file, ref_path, filter_extra.replace(" ", "\x00"), merge_extra.replace(" ", "\x00"), CONSENSUS_EXTRA.replace(" ", "\x00"),
            " --residentBytes %d" % resident if resident else ""))
    _compare(_snapshot(work, dirs, remove=False), want)
    st = hot_path.hot_path_batch.last_stats
    total = sum(len(p) for p in piles)
    assert st["file_bytes"] == total
    if not partial:
        assert st["h2d_bytes"] == total and st["resident_files"] == len(piles)     # every pileup crossed the host link exactly once
    else:
        assert 0 < st["resident_files"] < len(piles) and st["h2d_bytes"] > total


@pytest.mark.parametrize("world", [2, 3, 8])
def test_hot_path_batch_sharded_over_ranks_writes_the_same_files(tmp_path, monkeypatch, world):
    """torchrun, one rank per GPU in production (RCCL); here all ranks on the one GPU of the test box with gloo moving the
    bytes: contiguous blocks of the sorted samples per rank, C1 / C2 / row bands between them — every file as the one-rank job
    writes it (which the test above compares with the separate subcommands).  Eight ranks (the driver's scaling run) over seven
    samples: one block per rank and a rank with nothing at all, one distance tile for eight ranks."""
    import socket
    import subprocess
    import sys
    work = tmp_path
    ref_path, dirs, dirs_file, piles = _outbreak_tree(work, n_samples=7)
    filter_extra = "--edge_length 100 --window_size 1000 125 15 --max_snp 3 2 1 --mode all"
    line = ("hot_path_batch -f %s %s --filterRegionsExtraParams=%s --callConsensusExtraParams=%s --varscanExtraParams=%s"
            % (dirs_file, ref_path, filter_extra.replace(" ", "\x00"), CONSENSUS_EXTRA.replace(" ", "\x00"), VARSCAN_EXTRA.replace(" ", "\x00")))
    monkeypatch.chdir(work)
    _run(line)
    want = _snapshot(work, dirs)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the host side's CPU budget: a node whose cgroup grants 16 CPUs (whatever the box shows), shared by the ranks of the launch
    cgroup, stats_dir = work / "cgroup", work / "stats"
    cgroup.mkdir()
    stats_dir.mkdir()
    (cgroup / "cpu.max").write_text("1600000 100000\n")
    env = dict(os.environ, SNPGPU_PIPELINE_ONE_GPU="1", MASTER_ADDR="127.0.0.1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""),
               SNPGPU_CGROUP_ROOT=str(cgroup), SNPGPU_HOT_PATH_STATS=str(stats_dir))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bin", "cfsan_snp_pipeline")] + [w.replace("\x00", " ") for w in line.split()] + ["-v", "0"]
    r = subprocess.run(cmd, cwd=str(work), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    _compare(_snapshot(work, dirs, remove=False), want)
    import json
    per_rank = [json.load(open(str(stats_dir / ("rank%d.json" % k)))) for k in range(world)]
    usable = min(16, len(os.sched_getaffinity(0)))
    for st in per_rank:
        assert st["local_world"] == world and st["usable_cores"] == usable and st["cpu_budget"]["budget"] == max(1, usable // world)
        assert st["readers"] <= st["cpu_budget"]["readers"] <= max(1, usable // world)
    assert sum(st["readers"] for st in per_rank) <= max(usable, world)        # 8 ranks on 16 CPUs: at most 2 readers each (r5: 8 each)


def test_hot_path_batch_in_an_rccl_group_of_one_writes_the_same_files(tmp_path, monkeypatch):
    """The box of the GPU tests has one GPU, and RCCL wants one GPU per rank — so the N > 1 job runs here over gloo (above).  What
    CAN run here is RCCL itself: a process group of one rank with SNPGPU_DIST_AT_WORLD_1=1 makes every collective call of the job
    (all_gather_object of names and errors, the variable-length all-gather of site keys, the all-gather of packed rows into the
    padded matrix, the all-to-all of distance tiles with its split lists, barriers) on device tensors through backend nccl."""
    import socket
    import subprocess
    import sys
    work = tmp_path
    ref_path, dirs, dirs_file, piles = _outbreak_tree(work, n_samples=7)
    filter_extra = "--edge_length 100 --window_size 1000 125 15 --max_snp 3 2 1 --mode all"
    line = ("hot_path_batch -f %s %s --filterRegionsExtraParams=%s --callConsensusExtraParams=%s --varscanExtraParams=%s"
            % (dirs_file, ref_path, filter_extra.replace(" ", "\x00"), CONSENSUS_EXTRA.replace(" ", "\x00"), VARSCAN_EXTRA.replace(" ", "\x00")))
    monkeypatch.chdir(work)
    _run(line)
    want = _snapshot(work, dirs)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "SNPGPU_PIPELINE_ONE_GPU"}
    env.update(SNPGPU_DIST_AT_WORLD_1="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               TORCH_DISTRIBUTED_DEBUG="DETAIL", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    probe = ("import torch.distributed as dist, sys\n"
             "from snp_pipeline_amd import cfsan_snp_pipeline as cli, hot_path\n"
             "seen = []\n"
             "init = dist.init_process_group\n"
             "def spy(backend=None, *a, **k):\n"
             "    seen.append(backend)\n"
             "    return init(backend, *a, **k)\n"
             "dist.init_process_group = spy\n"
             "cli.run_command_from_args(cli.parse_argument_list(sys.argv[1:]))\n"
             "assert seen == ['nccl'], seen\n"
             "print('backend', seen[0])\n")
    cmd = [sys.executable, "-c", probe] + [w.replace("\x00", " ") for w in line.split()] + ["-v", "0"]
    r = subprocess.run(cmd, cwd=str(work), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "backend nccl" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    _compare(_snapshot(work, dirs, remove=False), want)
    # ... and with the exchanges left to torch.distributed (SNPGPU_COMM=torch) instead of the library's own communicator
    r = subprocess.run(cmd, cwd=str(work), env=dict(env, SNPGPU_COMM="torch", MASTER_PORT=str(port + 1 if port < 65000 else port - 1)),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "backend nccl" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    _compare(_snapshot(work, dirs, remove=False), want)


def _plain_tree(work, n_samples, genome_len=3000, variants=()):
    """Samples whose reads all show the reference, except at `variants` [(sample, pos, alt)]."""
    import random
    rng = random.Random(3)
    ref = "".join(rng.choice("ACGT") for _ in range(genome_len))
    ref_path = work / "reference" / "ref.fasta"
    ref_path.parent.mkdir()
    ref_path.write_text(_fasta("ctg1", ref))
    old = time.time() - 1000
    os.utime(str(ref_path), (old, old))
    dirs, piles = [], []
    for s in range(n_samples):
        alts = {p: a for (k, p, a) in variants if k == s}
        lines = []
        for pos in range(1, genome_len + 1):
            if pos in alts:
                bases = (alts[pos] + alts[pos].lower()) * 6
            else:
                bases = ".," * 6
            lines.append("ctg1\t%d\t%s\t12\t%s\t%s\n" % (pos, ref[pos - 1], bases, "I" * 12))
        data = "".join(lines).encode()
        sdir = work / "samples" / ("s%02d" % s)
        sdir.mkdir(parents=True)
        bam = sdir / "reads.sorted.deduped.indelrealigned.bam"
        bam.write_bytes(b"placeholder")
        os.utime(str(bam), (old, old))
        (sdir / "reads.all.pileup").write_bytes(data)
        dirs.append(str(sdir))
        piles.append(data)
    dirs_file = str(work / "sampleDirectories.txt")
    with open(dirs_file, "w") as f:
        f.write("\n".join(dirs) + "\n")
    return str(ref_path), dirs, dirs_file, piles


@pytest.mark.parametrize("case", ["no_variant_anywhere", "one_sample", "one_sample_without_variants"])
def test_hot_path_batch_on_the_smallest_jobs(tmp_path, monkeypatch, case):
    """Empty site lists (no sample differs from the reference) and a job of one sample: every file as the separate steps
    write it (empty snplists, header-only consensus files, TSVs of one row)."""
    work = tmp_path
    n = 3 if case == "no_variant_anywhere" else 1
    variants = [(0, 500, "A"), (0, 1500, "C"), (0, 2500, "G")] if case == "one_sample" else []
    ref_path, dirs, dirs_file, piles = _plain_tree(work, n, variants=variants)
    # (a variant allele that equals the reference base would be no variant: make them differ)
    ref_seq = "".join(open(ref_path).read().split("\n")[1:])
    variants = [(s, p, a if ref_seq[p - 1] != a else "T" if a != "T" else "A") for s, p, a in variants]
    if variants:
        import shutil
        shutil.rmtree(str(work / "samples")), shutil.rmtree(str(work / "reference"))
        ref_path, dirs, dirs_file, piles = _plain_tree(work, n, variants=variants)
    monkeypatch.setenv("VarscanMpileup2snp_ExtraParams", VARSCAN_EXTRA)
    monkeypatch.chdir(work)
    filter_extra = "--edge_length 100 --window_size 1000 --max_snp 3 --mode all"
    _separate_steps(work, ref_path, dirs, dirs_file, filter_extra, "")
    want = _snapshot(work, dirs)
    assert len(want["snplist.txt"].splitlines()) == (3 if case == "one_sample" else 0)
    _run("hot_path_batch -f %s %s --filterRegionsExtraParams=%s --callConsensusExtraParams=%s"
         % (dirs_file, ref_path, filter_extra.replace(" ", "\x00"), CONSENSUS_EXTRA.replace(" ", "\x00")))
    _compare(_snapshot(work, dirs, remove=False), want)


def test_hot_path_batch_goes_on_without_the_samples_that_fail(tmp_path, monkeypatch):
    """StopOnSampleError=false (run.py's default for the job arrays): a sample without a pileup and one whose pileup has a
    malformed position (site calling takes that column as text, call_consensus raises ValueError for it) are reported in the
    error log, and every file of the other samples and every top-level file comes out as the separate steps write them in the
    same situation."""
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    work = tmp_path
    ref_path, dirs, dirs_file, piles = _outbreak_tree(work, n_samples=6)
    os.remove(os.path.join(dirs[1], "reads.all.pileup"))
    bad = os.path.join(dirs[4], "reads.all.pileup")
    lines = open(bad, "rb").read().split(b"\n")
    f = lines[4000].split(b"\t")
    f[1] = b"12x"                                               # int(position) raises ValueError in the reference
    lines[4000] = b"\t".join(f)
    open(bad, "wb").write(b"\n".join(lines))
    log = work / "error.log"
    monkeypatch.setenv("StopOnSampleError", "false")
    monkeypatch.setenv("errorOutputFile", str(log))
    monkeypatch.setenv("VarscanMpileup2snp_ExtraParams", VARSCAN_EXTRA)
    monkeypatch.chdir(work)

    def run(line):                                              # 98 = "a sample failed, the others went on"
        args = cli.parse_argument_list([w.replace("\x00", " ") for w in line.split()])
        args.verbose = 0
        try:
            return cli.run_command_from_args(args)
        except SystemExit as e:
            assert e.code == 98, (line, e.code)
            return 98
        except (ValueError, IndexError):                         # what the sample exception hook turns into exit code 98
            return 98

    good = [d for i, d in enumerate(dirs) if i not in (1, 4)]
    filter_extra = "--edge_length 100 --window_size 1000 125 15 --max_snp 3 2 1 --mode all"
    for sdir in dirs:
        if sdir != dirs[1]:                                      # (without a pileup call_sites would start samtools, as the reference does)
            run("call_sites %s %s" % (ref_path, sdir))
    # (VarScan takes the position column as text: the malformed sample gets its var.flt.vcf and fails at call_consensus)
    assert os.path.getsize(os.path.join(dirs[4], "var.flt.vcf")) > 0
    run("filter_regions -f -n var.flt.vcf %s %s %s" % (dirs_file, ref_path, filter_extra))
    run("merge_sites -f -n var.flt.vcf -o %s/snplist.txt %s %s.OrigVCF.filtered" % (work, dirs_file, dirs_file))
    run("merge_sites -f -n var.flt_preserved.vcf -o %s/snplist_preserved.txt %s %s.PresVCF.filtered" % (work, dirs_file, dirs_file))
    for sdir in good:
        run("call_consensus -f -l %s/snplist.txt -o %s/consensus.fasta --vcfRefName ref.fasta %s --vcfFileName consensus.vcf %s/reads.all.pileup"
            % (work, sdir, CONSENSUS_EXTRA, sdir))
        run("call_consensus -f -l %s/snplist_preserved.txt -o %s/consensus_preserved.fasta -e %s/var.flt_removed.vcf --vcfRefName ref.fasta %s "
            "--vcfFileName consensus_preserved.vcf %s/reads.all.pileup" % (work, sdir, sdir, CONSENSUS_EXTRA, sdir))
    for suffix, flt in (("", "OrigVCF"), ("_preserved", "PresVCF")):
        run("snp_matrix -f -c consensus%s.fasta -o %s/snpma%s.fasta %s.%s.filtered" % (suffix, work, suffix, dirs_file, flt))
        run("snp_reference -f -l %s/snplist%s.txt -o %s/referenceSNP%s.fasta %s" % (work, suffix, work, suffix, ref_path))
        run("distance -f -p %s/snp_distance_pairwise%s.tsv -m %s/snp_distance_matrix%s.tsv %s/snpma%s.fasta" % (work, suffix, work, suffix, work, suffix))
    var_files = ("var.flt.vcf", "var.flt_preserved.vcf", "var.flt_removed.vcf")
    want_bad = {n: open(os.path.join(dirs[4], n), "rb").read() for n in var_files}
    want = _snapshot(work, good)
    assert want["snpma.fasta"].count(b">") == len(good)
    for sdir in (dirs[1], dirs[4]):                              # whatever the separate steps left for the failed samples goes away
        for name in PER_SAMPLE:
            if os.path.exists(os.path.join(sdir, name)):
                os.remove(os.path.join(sdir, name))
    log.write_text("")
    rc = run("hot_path_batch -f %s %s --filterRegionsExtraParams=%s --callConsensusExtraParams=%s"
             % (dirs_file, ref_path, filter_extra.replace(" ", "\x00"), CONSENSUS_EXTRA.replace(" ", "\x00")))
    assert rc in (0, 98)
    _compare(_snapshot(work, good, remove=False), want)
    assert {n: open(os.path.join(dirs[4], n), "rb").read() for n in var_files} == want_bad
    text = log.read_text()
    assert os.path.basename(dirs[1]) in text and os.path.basename(dirs[4]) in text
    assert not os.path.exists(os.path.join(dirs[4], "consensus.fasta"))


def test_hot_path_batch_records_the_collect_metrics_by_products(tmp_path, monkeypatch):
    """call_consensus --amdMetricsRefFasta (through CallConsensus_ExtraParams) records avePileupDepth, missingPos and
    missingPosPreserved in each sample's metrics file (collect_metrics.py:109-128, 325-340 reuse them); the one job writes the
    same files as the two call_consensus arrays."""
    work = tmp_path
    ref_path, dirs, dirs_file, piles = _outbreak_tree(work, n_samples=5)
    monkeypatch.setenv("VarscanMpileup2snp_ExtraParams", VARSCAN_EXTRA)
    monkeypatch.chdir(work)
    filter_extra = "--edge_length 100 --window_size 1000 125 15 --max_snp 3 2 1 --mode all"
    extra = CONSENSUS_EXTRA + " --vcfFailedSnpGt 1 --vcfPreserveRefCase --amdMetricsRefFasta " + ref_path    # (and the two VCF layout options)
    _separate_steps(work, ref_path, dirs, dirs_file, filter_extra, "", consensus_extra=extra)
    want = _snapshot(work, dirs)
    want_metrics = {}
    for sdir in dirs:
        path = os.path.join(sdir, "metrics")
        want_metrics[sdir] = open(path).read()
        assert "avePileupDepth=" in want_metrics[sdir] and "missingPos=" in want_metrics[sdir] and "missingPosPreserved=" in want_metrics[sdir]
        os.remove(path)
    _run("hot_path_batch -f %s %s --filterRegionsExtraParams=%s --callConsensusExtraParams=%s"
         % (dirs_file, ref_path, filter_extra.replace(" ", "\x00"), extra.replace(" ", "\x00")))
    _compare(_snapshot(work, dirs, remove=False), want)
    for sdir in dirs:
        assert open(os.path.join(sdir, "metrics")).read() == want_metrics[sdir], sdir


@pytest.mark.parametrize("tree, line_ends, filter_extra, merge_extra, consensus_extra, varscan_extra", [
    # --maxsnps takes samples out of snplist.txt for their var.flt.vcf and leaves them in snplist_preserved.txt for their shorter
    # var.flt_preserved.vcf: the second list is no subset of the first
    (dict(seed=681771, n_samples=3, genome_len=12000), {"iso01": "crlf"}, "--edge_length 500 --window_size 500 --max_snp 2 --mode all", "--maxsnps 10",
     "-q 10 -c 0.51 -D 1 -d 0 -b 0.0", VARSCAN_EXTRA),
    # a pileup that repeats positions: a consensus.vcf row for every matching line (the job hands that sample's VCF files to the
    # per-sample command)
    (dict(seed=681455, n_samples=2, genome_len=6000), {"iso01": "repeats"}, "--edge_length 1 --window_size 1000 125 15 --max_snp 3 2 1 --mode each", "",
     CONSENSUS_EXTRA, "--min-var-freq 0.5 --min-reads2 3 --p-value 1e-6 --strand-filter 0"),
    # --vcfAllPos among the call_consensus options: a VCF row for every pileup line, written by the per-sample command at the end
    (dict(seed=11, n_samples=3, genome_len=2500), {"iso02": "crlf"}, "--edge_length 100 --window_size 1000 125 15 --max_snp 3 2 1 --mode all", "",
     "-q 10 -c 0.51 -D 1 -d 0 -b 0.0 --vcfAllPos", VARSCAN_EXTRA),
])
def test_hot_path_batch_on_jobs_the_fuzz_campaign_found(tmp_path, monkeypatch, tree, line_ends, filter_extra, merge_extra, consensus_extra, varscan_extra):
    """Jobs tools/fuzz_jobs.py stopped at (and one with --vcfAllPos, which the job used to refuse): every output file of the one job
    equals the separate subcommands'."""
    work = tmp_path
    ref_path, dirs, dirs_file, piles = _outbreak_tree(work, **tree)
    for i, sdir in enumerate(dirs):
        variant = line_ends.get(os.path.basename(sdir))
        if variant:
            with open(os.path.join(sdir, "reads.all.pileup"), "wb") as f:
                f.write(fuzz.with_line_ends(piles[i], variant, tree["seed"] + i))
    monkeypatch.setenv("VarscanMpileup2snp_ExtraParams", varscan_extra)
    monkeypatch.chdir(work)
    _separate_steps(work, ref_path, dirs, dirs_file, filter_extra, merge_extra, consensus_extra)
    want = _snapshot(work, dirs)
    if merge_extra:
        first, second = set(want["snplist.txt"].splitlines()), set(ln.split(b"\t")[0] + b"\t" + ln.split(b"\t")[1] for ln in want["snplist_preserved.txt"].splitlines())
        assert second - set(ln.split(b"\t")[0] + b"\t" + ln.split(b"\t")[1] for ln in first)       # positions only the preserved list has
    _run("hot_path_batch -f %s %s --filterRegionsExtraParams=%s --mergeSitesExtraParams=%s --callConsensusExtraParams=%s"
         % (dirs_file, ref_path, filter_extra.replace(" ", "\x00"), merge_extra.replace(" ", "\x00"), consensus_extra.replace(" ", "\x00")))
    _compare(_snapshot(work, dirs, remove=False), want)


def test_hot_path_batch_with_many_symbols_at_a_position_of_a_sample_that_is_not_resident(tmp_path, monkeypatch):
    """A listed position with more than eight distinct symbols needs the spill records of ONE library call for its consensus.vcf
    row; a group that is only partly resident has no such call, so the job lets the per-sample command write that sample's two
    VCF files at the end — every file as the separate steps write it (until this round the job reported the sample instead)."""
    work = tmp_path
    ref_path, dirs, dirs_file, piles = _outbreak_tree(work)
    monkeypatch.setenv("VarscanMpileup2snp_ExtraParams", VARSCAN_EXTRA)
    monkeypatch.chdir(work)
    filter_extra = "--edge_length 100 --window_size 1000 125 15 --max_snp 3 2 1 --mode all"
    # a position several samples carry, so that it stays on the list whatever the rewritten line does to its own sample's calls
    for sdir in dirs:
        _run("call_sites %s %s" % (ref_path, sdir))
    _run("merge_sites -f -n var.flt.vcf -o %s/probe.txt %s %s.probe" % (work, dirs_file, dirs_file))
    chrom, pos = next((f[0], int(f[1])) for f in (ln.split("\t") for ln in open(str(work / "probe.txt"))) if int(f[2]) >= 3)
    for sdir in (dirs[-1], dirs[0]):                              # the last sample is streamed twice in the run below, the first one is resident
        path = os.path.join(sdir, "reads.all.pileup")
        lines = open(path, "rb").read().split(b"\n")
        k = next(i for i, ln in enumerate(lines) if ln.startswith(b"%s\t%d\t" % (chrom.encode(), pos)))
        f = lines[k].split(b"\t")
        f[3], f[4], f[5] = b"24", b"ACGTNRYKMSWBacgtnrykmswb", b"I" * 24
        lines[k] = b"\t".join(f)
        open(path, "wb").write(b"\n".join(lines))
    _separate_steps(work, ref_path, dirs, dirs_file, filter_extra, "")
    want = _snapshot(work, dirs)
    row = next(ln for ln in want[os.path.join(os.path.basename(dirs[-1]), "consensus.vcf")].split(b"\n") if ln.startswith(b"%s\t%d\t" % (chrom.encode(), pos)))
    assert row.split(b"\t")[4].count(b",") >= 9                   # ten or more ALT alleles in that row
    resident = int(2.5 * max(len(p) for p in piles))
    _run("hot_path_batch -f %s %s --filterRegionsExtraParams=%s --callConsensusExtraParams=%s --residentBytes %d"
         % (dirs_file, ref_path, filter_extra.replace(" ", "\x00"), CONSENSUS_EXTRA.replace(" ", "\x00"), resident))
    _compare(_snapshot(work, dirs, remove=False), want)
    from snp_pipeline_amd import hot_path
    assert 0 < hot_path.hot_path_batch.last_stats["resident_files"] < len(dirs)
    # and with everything resident (the group's own spill records)
    for name in list(want):
        p = os.path.join(str(work), name) if name in TOP_LEVEL else os.path.join(str(work), "samples", name)
        os.remove(p)
    _run("hot_path_batch -f %s %s --filterRegionsExtraParams=%s --callConsensusExtraParams=%s"
         % (dirs_file, ref_path, filter_extra.replace(" ", "\x00"), CONSENSUS_EXTRA.replace(" ", "\x00")))
    _compare(_snapshot(work, dirs, remove=False), want)


# ---------------------------------------------------------------------------------------------------------------------------
# Site calling mode ``existing``: a tree whose var.flt.vcf files were written by somebody else — here the reference's own
# bundled files, i.e. real VarScan output — goes through the job with those files as INPUTS.  They must stay byte for byte (and
# keep their modification times), and everything downstream of them that the reference ships must come out as bundled: the split
# VCF files, both snplists, both referenceSNP files.  The consensus side is checked against the restatement on the pileups the
# test makes (the reference ships no pileup).
# ---------------------------------------------------------------------------------------------------------------------------
def _foreign_tree(tmp_path, fixture_trees, ds):
    import lzma
    import shutil
    from tests.conftest import GOLD
    from snp_pipeline_amd import utils
    root, meta = fixture_trees[ds]
    work = tmp_path / "work"
    (work / "samples").mkdir(parents=True)
    ref_path = str(work / "reference.fasta")
    if ds == "lambdaVirus":
        shutil.copy(os.path.join(GOLD, "fixtures", "lambdaVirus", "lambda_virus.fasta"), ref_path)
    else:
        with lzma.open(os.path.join(GOLD, "fixtures", "listeria", "CFSAN023463.HGAP.draft.fasta.xz")) as f, open(ref_path, "wb") as out:
            out.write(f.read())
    os.utime(ref_path, (time.time() - 2000, time.time() - 2000))     # older than the placeholder BAMs and the pileups: nothing upstream is stale
    refs = dict(utils.fasta_records_ascii(ref_path))
    refs = {k: v.decode("ascii") for k, v in refs.items()}
    contig_order = list(refs)
    names = sorted(os.listdir(os.path.join(root, "samples")))
    per_sample = {}
    for name in names:
        _, _, sites = utils.read_vcf_sites(os.path.join(root, "samples", name, "var.flt.vcf"))
        per_sample[name] = sites
    union = {}
    for sites in per_sample.values():
        for c, p in sites:
            union.setdefault(c, set()).add(p)
    layout = [(c, sorted(union[c])) for c in contig_order if c in union]
    dirs, piles = [], {}
    old = time.time() - 1000
    for k, name in enumerate(names):
        sdir = work / "samples" / name
        sdir.mkdir()
        piles[name] = fuzz.pileup_at_positions(100 + k, layout, refs)
        bam = sdir / "reads.sorted.deduped.indelrealigned.bam"
        bam.write_bytes(b"placeholder: the pileup is newer, so samtools is not run (call_sites.py:70-72)")
        os.utime(str(bam), (old - 10, old - 10))
        (sdir / "reads.all.pileup").write_bytes(piles[name])
        os.utime(str(sdir / "reads.all.pileup"), (old, old))
        shutil.copy(os.path.join(root, "samples", name, "var.flt.vcf"), str(sdir / "var.flt.vcf"))
        dirs.append(str(sdir))
    dirs_file = str(work / "sampleDirectories.txt")
    with open(dirs_file, "w") as f:
        f.write("\n".join(reversed(dirs)) + "\n")
    return root, str(work), ref_path, names, dirs, dirs_file, piles


@pytest.mark.parametrize("ds", ["lambdaVirus", "listeria"])
def test_hot_path_batch_with_foreign_var_flt_vcf_files(tmp_path, fixture_trees, monkeypatch, ds):
    import filecmp
    from snp_pipeline_amd import hot_path
    from snp_pipeline_amd import utils
    root, work, ref_path, names, dirs, dirs_file, piles = _foreign_tree(tmp_path, fixture_trees, ds)
    monkeypatch.chdir(work)
    monkeypatch.delenv("SNPGPU_SITE_CALLING", raising=False)
    before = {d: (open(os.path.join(d, "var.flt.vcf"), "rb").read(), os.stat(os.path.join(d, "var.flt.vcf")).st_mtime_ns) for d in dirs}
    _run("hot_path_batch -f --siteCalling existing %s %s --filterRegionsExtraParams=%s --callConsensusExtraParams=%s"
         % (dirs_file, ref_path, "--edge_length\x00500\x00--window_size\x001000\x00125\x0015\x00--max_snp\x003\x002\x001\x00--mode\x00all",
            CONSENSUS_EXTRA.replace(" ", "\x00")))
    assert hot_path.hot_path_batch.last_stats["site_calling"] == "existing"
    # 1. nobody touched the foreign files
    for d in dirs:
        path = os.path.join(d, "var.flt.vcf")
        assert (open(path, "rb").read(), os.stat(path).st_mtime_ns) == before[d], d
    # 2. what the reference ships downstream of them
    n_split = 0
    for name, d in zip(names, dirs):
        for fname in ("var.flt_preserved.vcf", "var.flt_removed.vcf"):
            if os.path.exists(os.path.join(root, "samples", name, fname)):       # (the listeria fixture ships var.flt.vcf and the snplists only)
                assert filecmp.cmp(os.path.join(d, fname), os.path.join(root, "samples", name, fname), shallow=False), (name, fname)
                n_split += 1
    assert n_split == (2 * len(names) if ds == "lambdaVirus" else 0)
    for fname in ("snplist.txt", "snplist_preserved.txt", "referenceSNP.fasta", "referenceSNP_preserved.fasta"):
        assert filecmp.cmp(os.path.join(work, fname), os.path.join(root, fname), shallow=False), fname
    # 3. the consensus side against the restatement on these pileups, both flows
    cprm = po.CallerParams(15, 0.9, 5, 2, 0.1)
    for suffix in ("", "_preserved"):
        snp_keys = [(c.encode(), p) for c, p in utils.read_snp_position_list(os.path.join(work, "snplist%s.txt" % suffix))]
        seqs = {}
        for name, d in zip(names, dirs):
            excluded = set()
            if suffix:
                _, _, rm = utils.read_vcf_sites(os.path.join(d, "var.flt_removed.vcf"))
                excluded = set((c.encode(), p) for c, p in rm)
            want, _ = po.call_consensus_sites(piles[name], snp_keys, excluded, cprm)
            seqs[name] = want.decode()
            assert open(os.path.join(d, "consensus%s.fasta" % suffix)).read() == _fasta(name, seqs[name]), (name, suffix)
        assert open(os.path.join(work, "snpma%s.fasta" % suffix)).read() == "".join(_fasta(n, seqs[n]) for n in names)
        ids, table = so.distance_tables(seqs)
        assert open(os.path.join(work, "snp_distance_pairwise%s.tsv" % suffix)).read() == so.pairwise_text(ids, table)
        assert open(os.path.join(work, "snp_distance_matrix%s.tsv" % suffix)).read() == so.matrix_text(ids, table)
    # 4. the batch call_sites in the same mode checks and writes nothing; a sample without the file is that sample's error
    _run("call_sites_batch --siteCalling existing %s %s" % (ref_path, dirs_file))


def test_site_calling_mode_existing_reports_a_sample_without_its_vcf(tmp_path, fixture_trees, monkeypatch, capfd):
    root, work, ref_path, names, dirs, dirs_file, piles = _foreign_tree(tmp_path, fixture_trees, "lambdaVirus")
    monkeypatch.chdir(work)
    monkeypatch.setenv("StopOnSampleError", "false")
    monkeypatch.setenv("SNPGPU_SITE_CALLING", "existing")        # the environment's spelling of --siteCalling
    os.remove(os.path.join(dirs[1], "var.flt.vcf"))
    late = time.time() + 100
    os.utime(os.path.join(dirs[2], "reads.all.pileup"), (late, late))    # its var.flt.vcf is now older than its pileup
    _run("hot_path_batch -f %s %s --filterRegionsExtraParams=%s --callConsensusExtraParams=%s"
         % (dirs_file, ref_path, "--edge_length\x00500", CONSENSUS_EXTRA.replace(" ", "\x00")))
    err = capfd.readouterr().err
    assert "var.flt.vcf does not exist" in err and "is older than" in err
    assert not os.path.exists(os.path.join(dirs[1], "var.flt.vcf"))
    for k in (0, 3):
        assert os.path.getsize(os.path.join(dirs[k], "consensus.fasta")) > 0
    got = open(os.path.join(work, "snpma.fasta")).read()
    assert got.count(">") == 2 and ">%s\n" % names[0] in got and ">%s\n" % names[3] in got


def test_a_rank_that_fails_takes_the_others_with_it_instead_of_leaving_them_in_a_collective(tmp_path):
    """one rank's stage raises (here: a split VCF of ONE of rank 1's samples cannot be written — a directory sits where
    the file should go) while its peer is healthy.  Between stages the ranks agree on failure: both leave, soon, with a non-zero
    exit code; the failing rank's own error reaches the log; nobody waits in the next all-gather until a timeout ends it."""
    import socket
    import subprocess
    import sys
    work = tmp_path
    ref_path, dirs, dirs_file, piles = _outbreak_tree(work, n_samples=6)
    os.makedirs(os.path.join(sorted(dirs)[-1], "var.flt_removed.vcf"))          # the last sample belongs to rank 1
    filter_extra = "--edge_length 100 --window_size 1000 125 15 --max_snp 3 2 1 --mode all"
    line = ("hot_path_batch -f --siteCalling device %s %s --filterRegionsExtraParams=%s --callConsensusExtraParams=%s --varscanExtraParams=%s"
            % (dirs_file, ref_path, filter_extra.replace(" ", "\x00"), CONSENSUS_EXTRA.replace(" ", "\x00"), VARSCAN_EXTRA.replace(" ", "\x00")))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    log = str(work / "error.log")
    env = dict(os.environ, SNPGPU_PIPELINE_ONE_GPU="1", MASTER_ADDR="127.0.0.1", errorOutputFile=log,
               PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bin", "cfsan_snp_pipeline")] + [w.replace("\x00", " ") for w in line.split()] + ["-v", "1"]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=str(work), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and time.time() - t0 < 120, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    text = open(log).read()
    assert "IsADirectoryError" in text and "hot_path_batch" in text
    assert "rank 0 stops after stage_consensus: rank 1 failed" in r.stdout, r.stdout[-2500:]


def test_the_stages_of_the_job_can_be_driven_one_at_a_time(tmp_path, monkeypatch):
    """hot_path_batch is five stage functions over one job-state object (round 3: one 680-line function): here the first two are
    run alone — ingest + site calling, then site union + region filter — and what they leave in the job and on disk is what
    call_sites / filter_regions / merge_sites x 2 write; then the rest, and the job's files are the separate steps' files."""
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    from snp_pipeline_amd import hot_path
    work = tmp_path
    ref_path, dirs, dirs_file, piles = _outbreak_tree(work, n_samples=5)
    monkeypatch.chdir(work)
    monkeypatch.setenv("VarscanMpileup2snp_ExtraParams", VARSCAN_EXTRA)
    filter_extra = "--edge_length 100 --window_size 1000 125 15 --max_snp 3 2 1 --mode all"
    _separate_steps(work, ref_path, dirs, dirs_file, filter_extra, "")
    want = _snapshot(work, dirs)
    args = cli.parse_argument_list(["hot_path_batch", "-f", "--siteCalling", "device", dirs_file, ref_path, "--filterRegionsExtraParams=" + filter_extra,
                                    "--callConsensusExtraParams=" + CONSENSUS_EXTRA, "--varscanExtraParams=" + VARSCAN_EXTRA])
    args.verbose = 0
    job = hot_path._Job(args, hot_path._Comm())
    job.open_device()
    try:
        job.run_stage(hot_path.stage_ingest_and_sites)
        assert all(s.ok and s.store_index >= 0 and len(s.sites[2]) > 5 for s in job.mine) and len(job.store) == len(dirs)
        for s in job.mine:
            assert open(os.path.join(s.dir, "var.flt.vcf"), "rb").read() == want[os.path.join(s.name, "var.flt.vcf")]
        job.run_stage(hot_path.stage_site_union_and_regions)
        for fu in job.split_files:
            fu.result()
        for name in ("snplist.txt", "snplist_preserved.txt", "sampleDirectories.txt.OrigVCF.filtered", "sampleDirectories.txt.PresVCF.filtered"):
            assert open(os.path.join(str(work), name), "rb").read() == want[name], name
        for s in job.mine:
            for name in ("var.flt_preserved.vcf", "var.flt_removed.vcf"):
                assert open(os.path.join(s.dir, name), "rb").read() == want[os.path.join(s.name, name)], (s.name, name)
        assert len(job.list1) >= len(job.list2) > 10 and not job.excluded1.any()
        for stage in hot_path.STAGES[2:]:
            job.run_stage(stage)
        assert job.row_ok.all() and job.flows.S1 == len(job.list1)
    finally:
        job.close_device()
    _compare(_snapshot(work, dirs, remove=False), want)

"""Streamed ingestion (csrc/stream.hip): pileup FILES -> consensus, scanned while they arrive.

The streamed path must give exactly what the resident path gives (which the other test files pin against the
reference-generated vectors and the oracle) for any chunking, any number of files in flight, and any file shape; plus the
oracle directly on multi-chunk files, the per-file error codes, the call_consensus_batch subcommand, --vcfAllPos, and
eight concurrent CLI processes sharing one device through the slot locks.
"""
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from oracle import fuzz
from oracle import pileup_oracle as po
from oracle import vcf_oracle as vo

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bin", "cfsan_snp_pipeline")


@pytest.fixture(scope="module")
def d():
    from tests.gpu_util import get_device
    return get_device()


def _resident(d, ss, datas, prm):
    """The same bytes through the resident batch entry point (device pointers)."""
    import torch
    d.use_torch_stream()
    n = len(ss)
    sizes = np.asarray([len(x) for x in datas], dtype=np.uint64)
    offs = np.zeros(len(datas), dtype=np.uint64)
    for i in range(1, len(datas)):
        offs[i] = offs[i - 1] + (int(sizes[i - 1]) + 255) // 256 * 256
    blob = np.zeros(int(offs[-1]) + int(sizes[-1]) + 64, dtype=np.uint8)
    for i, x in enumerate(datas):
        blob[int(offs[i]):int(offs[i]) + len(x)] = np.frombuffer(x, dtype=np.uint8)
    t = torch.from_numpy(blob).cuda()
    bases = torch.zeros((len(datas), n), dtype=torch.uint8, device="cuda")
    filt = torch.zeros((len(datas), n), dtype=torch.uint8, device="cuda")
    status = torch.zeros((len(datas), 4), dtype=torch.int64, device="cuda")
    d.call_consensus_batch_dev(ss, t.data_ptr(), offs, prm, bases.data_ptr(), filt.data_ptr(), status.data_ptr(), sizes=sizes)
    torch.cuda.synchronize()
    return bases.cpu().numpy(), filt.cpu().numpy(), status.cpu().numpy().view(np.uint64)


def test_files_any_chunking_equals_resident_and_oracle(d, tmp_path):
    from snp_pipeline_amd import _lib as L
    from snp_pipeline_amd import device as dev
    big, _, sites = fuzz.synth_pileup(71, genome_len=40000, n_sites=500)             # ~3.4 MB: ~50 chunks of 64 KiB
    multi, _, sites2 = fuzz.synth_pileup(72, genome_len=9000, n_sites=200, contigs=("synth_chr1", "ctgB", "a"))
    crlf = big[:400000].replace(b"\n", b"\r\n")
    cut = big[:3 * 65536]                                                              # ends exactly at a chunk boundary, unterminated
    keep = cut.rindex(b"\n", 0, len(cut) - 40) + 1
    tail = b"synth_chr1\t39999\tA\t0\t*\t"
    cut = cut[:keep] + tail + b"x" * (3 * 65536 - keep - len(tail))
    assert len(cut) == 3 * 65536 and not cut.endswith(b"\n")
    datas = [big, b"", multi, b"synth_chr1\t700\tA\t3\t...\tIII", crlf, cut, big[:65536 + 4096 + 100], multi + big]
    paths = []
    for i, x in enumerate(datas):
        p = tmp_path / ("f%d.pileup" % i)
        p.write_bytes(x)
        paths.append(str(p))
    keys = sorted(set(sites + sites2 + [(b"synth_chr1", 700), (b"synth_chr1", 39999), (b"zz", 5)]))
    flags = [L.SITE_IN_SNPLIST | (L.SITE_EXCLUDED if i % 9 == 0 else 0) for i in range(len(keys))]
    ss = d.siteset(keys, flags)
    p = po.CallerParams(0, 0.6, 3, 0, 0.0)
    prm = dev.make_params(0, 0.6, 3, 0, 0.0)
    want_b, want_f, want_s = _resident(d, ss, datas, prm)
    for kw in (dict(chunk_bytes=65536, n_staging=3, n_readers=2, n_slots=2), dict(chunk_bytes=65536, n_staging=1, n_readers=1, n_slots=1),
               dict(chunk_bytes=1 << 20, n_staging=5, n_readers=4, n_slots=3), dict()):
        results, rcs, stats = d.call_consensus_files(ss, paths, prm, want_counts=True, want_line_offsets=True, want_depth_sum=False, **kw)
        assert list(rcs) == [0] * len(paths)
        assert stats.bytes == sum(len(x) for x in datas)
        for i, r in enumerate(results):
            assert bytes(r.bases) == bytes(want_b[i]) and bytes(r.filters) == bytes(want_f[i]), (kw, i)
            assert r.status.tolist()[:3] == want_s[i].tolist()[:3], (kw, i)
            assert r.n_lines == sum(1 for _ in po.iter_lines(datas[i]))
            # the line offsets point at the start of a line of the right position
            for slot in np.nonzero(r.line_offsets)[0][:50]:
                off = int(r.line_offsets[slot]) - 1
                f = po.split_fields(datas[i][off:off + 200].split(b"\n")[0].split(b"\r")[0])
                assert (f[0], int(f[1])) == ss.key_tuples()[slot]
        # the lane-per-site path (no counts) and the depth-sum variant of the scan
        r2, rc2, _ = d.call_consensus_files(ss, paths, prm, want_depth_sum=True, **kw)
        for i, r in enumerate(r2):
            assert bytes(r.bases) == bytes(want_b[i]) and bytes(r.filters) == bytes(want_f[i])
            assert r.depth_sum == sum(int(f[3]) for _, ln in po.iter_lines(datas[i]) for f in [ln.split()] if len(f) > 3)
    # ... and the oracle itself on the multi-chunk files
    excl = {k for k, fl in zip(ss.key_tuples(), ss.flags) if fl & L.SITE_EXCLUDED}
    results, _, _ = d.call_consensus_files(ss, paths[:5], prm, chunk_bytes=65536, n_staging=4, n_readers=3)
    for i in range(5):
        want, _ = po.call_consensus_sites(datas[i], ss.key_tuples(), excl, p)
        assert bytes(results[i].bases) == want


def test_files_per_file_errors(d, tmp_path):
    from snp_pipeline_amd import _lib as L
    from snp_pipeline_amd import device as dev
    good, _, sites = fuzz.synth_pileup(73, genome_len=3000, n_sites=60)
    bad = good[:20000] + b"oops\n" + good[20000:]                                      # a one-field line: ValueError in the reference
    paths = []
    for i, x in enumerate([good, bad, good]):
        p = tmp_path / ("g%d.pileup" % i)
        p.write_bytes(x)
        paths.append(str(p))
    paths.insert(1, str(tmp_path / "absent.pileup"))
    ss = d.siteset(sites, [L.SITE_IN_SNPLIST] * len(sites))
    prm = dev.make_params(0, 0.6, 3, 0, 0.0)
    results, rcs, _ = d.call_consensus_files(ss, paths, prm)
    assert list(rcs) == [0, L.E_IO, L.E_PILEUP, 0]
    want, _ = po.call_consensus_sites(good, ss.key_tuples(), set(), po.CallerParams(0, 0.6, 3, 0, 0.0))
    assert bytes(results[0].bases) == want and bytes(results[3].bases) == want
    with pytest.raises(dev.PileupIOError):
        d.raise_file_status(paths[1], rcs[1], results[1])
    with pytest.raises(dev.PileupFormatError) as ei:
        d.raise_file_status(paths[2], rcs[2], results[2])
    assert ei.value.reference_exception is ValueError


def _write_sample(work, name, data):
    sdir = work / name
    sdir.mkdir()
    (sdir / "reads.all.pileup").write_bytes(data)
    return sdir


def _fasta(name, cons):
    return ">%s\n" % name + "".join(cons[i:i + 60] + "\n" for i in range(0, len(cons), 60))


def test_eight_concurrent_cli_processes_on_one_device(tmp_path):
    """run.py:709-710 starts up to max_cpu_cores call_consensus processes at once; none knows about the others.  Eight of
    them against one device, at most three contexts at a time (slot locks), all results right."""
    _, _, sites = fuzz.synth_pileup(80, genome_len=30000, n_sites=400)
    with open(str(tmp_path / "snplist.txt"), "w") as f:
        for c, p in sites:
            f.write("%s\t%d\t1\ts\n" % (c.decode(), p))
    datas = {}
    for i in range(8):
        datas["s%d" % i] = fuzz.synth_pileup(80 + i, genome_len=30000, n_sites=400)[0]
        _write_sample(tmp_path, "s%d" % i, datas["s%d" % i])
    env = dict(os.environ, SNPGPU_MAX_PROCS_PER_DEVICE="3", SNPGPU_LOCK_DIR=str(tmp_path / "locks"))
    env.pop("SNPGPU_DEVICE", None)
    env.pop("LOCAL_RANK", None)
    procs = [subprocess.Popen([sys.executable, EXE, "call_consensus", "-v", "0", "-l", str(tmp_path / "snplist.txt"),
                               "-o", str(tmp_path / name / "consensus.fasta"), "--minConsDpth", "3", "--vcfFileName", "consensus.vcf",
                               str(tmp_path / name / "reads.all.pileup")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for name in sorted(datas)]
    for pr in procs:
        out, err = pr.communicate(timeout=600)
        assert pr.returncode == 0, err.decode()[-2000:]
    for name, data in datas.items():
        want, _ = po.call_consensus_sites(data, sites, set(), po.CallerParams(0, 0.6, 3, 0, 0.0))
        assert (tmp_path / name / "consensus.fasta").read_text() == _fasta(name, want.decode())
    locks = sorted(os.listdir(str(tmp_path / "locks")))
    assert locks and set(locks) <= {"dev0.slot%d" % j for j in range(3)}, locks


def test_call_consensus_batch_equals_per_sample_cli(tmp_path):
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    _, _, sites = fuzz.synth_pileup(90, genome_len=8000, n_sites=150)
    names = ["b%d" % i for i in range(5)]
    for i, name in enumerate(names):
        data = fuzz.synth_pileup(90 + i, genome_len=8000, n_sites=150)[0]
        sdir = _write_sample(tmp_path, name, data)
        with open(str(sdir / "var.flt_removed.vcf"), "w") as f:
            f.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n")
            for c, p in sites[i::7]:
                f.write("%s\t%d\t.\tA\tC\t.\tPASS\t.\tGT\t1/1\n" % (c.decode(), p))
    with open(str(tmp_path / "snplist.txt"), "w") as f:
        for c, p in sites:
            f.write("%s\t%d\t1\ts\n" % (c.decode(), p))
    with open(str(tmp_path / "dirs.txt"), "w") as f:
        for name in names:
            f.write("%s\n" % (tmp_path / name))
    common = "-v 0 -f -l %s/snplist.txt --minConsDpth 3 --vcfRefName ref.fa" % tmp_path
    for tag, excl in (("plain", ""), ("pres", " -e var.flt_removed.vcf")):
        assert cli.run_command_from_line("call_consensus_batch %s -o batch_%s.fasta --vcfFileName batch_%s.vcf%s %s/dirs.txt"
                                         % (common, tag, tag, excl, tmp_path)) == 0
        for name in names:
            sdir = tmp_path / name
            e1 = (" -e %s/var.flt_removed.vcf" % sdir) if excl else ""
            assert cli.run_command_from_line("call_consensus %s -o %s/one_%s.fasta --vcfFileName one_%s.vcf%s %s/reads.all.pileup"
                                             % (common, sdir, tag, tag, e1, sdir)) == 0
            assert (sdir / ("batch_%s.fasta" % tag)).read_text() == (sdir / ("one_%s.fasta" % tag)).read_text()
            assert (sdir / ("batch_%s.vcf" % tag)).read_text() == (sdir / ("one_%s.vcf" % tag)).read_text()
            assert len((sdir / ("batch_%s.fasta" % tag)).read_text()) > len(sites)


def test_vcf_all_pos_rows_for_every_line(tmp_path, monkeypatch):
    """--vcfAllPos (call_consensus.py:148-151): one VCF row per pileup line, listed or not; Region for excluded positions;
    the FASTA is unchanged.  A malformed line anywhere makes the reference raise: same exception type here."""
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    data, _, sites = fuzz.synth_pileup(95, genome_len=2500, n_sites=60, contigs=("ctgB", "ctgA"))
    data += b"ctgA\t999999\tN\t0\t*\t*\nctgZ\t5\tg\t4\t.,.^F,\tIIII\n"
    sdir = _write_sample(tmp_path, "sampleA", data)
    excl = sites[::5] + [(b"ctgZ", 5)]
    with open(str(tmp_path / "snplist.txt"), "w") as f:
        for c, p in sites:
            f.write("%s\t%d\t1\ts\n" % (c.decode(), p))
    with open(str(sdir / "excl.vcf"), "w") as f:
        f.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n")
        for c, p in excl:
            f.write("%s\t%d\t.\tA\tC\t.\tPASS\t.\tGT\t1/1\n" % (c.decode(), p))
    params = po.CallerParams(15, 0.9, 5, 2, 0.1)
    line = ("call_consensus -v 0 -f -l %s/snplist.txt -o %s/consensus.fasta -e %s/excl.vcf -q 15 -c 0.9 -D 5 -d 2 -b 0.1 "
            "--vcfFileName all.vcf --vcfAllPos --vcfPreserveRefCase %s/reads.all.pileup" % (tmp_path, sdir, sdir, sdir))
    assert cli.run_command_from_line(line) == 0
    want, _ = po.call_consensus_sites(data, sites, set(excl), params)
    assert (sdir / "consensus.fasta").read_text() == _fasta("sampleA", want.decode())
    names = po.filter_names(params)
    rows = []
    for _, ln in po.iter_lines(data):
        rec = po.parse_record(po.split_fields(ln), params.min_base_quality)
        base, mask = po.call_record(rec, params)
        if (rec.chrom, rec.position) in set(excl):
            mask |= 32
        failed = [names[i] for i in range(6) if mask >> i & 1] or None
        rows.append(vo.vcf_row(rec, failed, ".", preserve_ref_case=True))
    got = [x for x in (sdir / "all.vcf").read_text().split("\n") if x and not x.startswith("#")]
    assert len(got) == data.count(b"\n") and got == rows
    # malformed lines at unlisted positions only matter with --vcfAllPos
    monkeypatch.delenv("errorOutputFile", raising=False)
    for junk, exc in ((b"ctgA\t999998\tA\n", IndexError), (b"ctgA\t999998\tA\tx\t.\tI\n", ValueError), (b"ctgA\t999998\tA\t2\t..\n", IndexError)):
        (sdir / "reads.all.pileup").write_bytes(data + junk)
        assert cli.run_command_from_line(line.replace(" --vcfAllPos", "")) == 0
        with pytest.raises(exc):
            cli.run_command_from_line(line)


def test_batch_ignores_lines_at_positions_only_other_samples_exclude_and_sees_superseded_lines(tmp_path, monkeypatch):
    """One site set serves the whole batch, other samples' exclude positions included.  A malformed line at a position that is on
    neither of a sample's own lists is no business of that sample (the reference builds no Record there): the batch and the
    per-sample command both finish.  And a listed position that comes twice with a malformed FIRST line ends the sample with the
    reference's exception class although the per-site result only knows the last line (found while reading
    tools/fuzz_campaign.py's first finding)."""
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    monkeypatch.setenv("errorOutputFile", str(tmp_path / "error.log"))
    monkeypatch.setenv("StopOnSampleError", "false")
    lines = [b"c1\t%d\tA\t4\t....\tIIII\n" % k for k in range(1, 200)]
    a = list(lines)
    a[49] = b"c1\t50\tA\t4\t....\n"                                 # position 50: only on sample B's exclude list
    b = list(lines)
    c = list(lines)
    c.insert(20, b"c1\t120\tA\t4\t....\n")                           # position 120 (listed) comes early without qualities, and again
    for name, body in (("A", a), ("B", b), ("C", c)):
        sdir = _write_sample(tmp_path, name, b"".join(body))
        with open(str(sdir / "excl.vcf"), "w") as f:
            f.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n")
            if name == "B":
                f.write("c1\t50\t.\tA\tC\t.\tPASS\t.\tGT\t1/1\n")
    with open(str(tmp_path / "snplist.txt"), "w") as f:
        for p in (10, 120, 150):
            f.write("c1\t%d\t1\ts\n" % p)
    with open(str(tmp_path / "dirs.txt"), "w") as f:
        f.write("".join("%s\n" % (tmp_path / n) for n in "ABC"))
    common = "-v 0 -f -l %s/snplist.txt -e excl.vcf" % tmp_path
    # (sample C fails; with StopOnSampleError=false that is a logged sample error and the batch goes on)
    assert cli.run_command_from_line("call_consensus_batch %s -o batch.fasta %s/dirs.txt" % (common, tmp_path)) == 0
    assert not (tmp_path / "C" / "batch.fasta").exists()
    log = (tmp_path / "error.log").read_text()
    assert "sample C" in log and "IndexError" in log and "sample A" not in log and "sample B" not in log
    for name in "AB":
        sdir = tmp_path / name
        assert cli.run_command_from_line("call_consensus -v 0 -f -l %s/snplist.txt -e %s/excl.vcf -o %s/one.fasta %s/reads.all.pileup"
                                         % (tmp_path, sdir, sdir, sdir)) == 0
        assert (sdir / "batch.fasta").read_text() == (sdir / "one.fasta").read_text() == ">%s\nAAA\n" % name
    sdir = tmp_path / "C"
    with pytest.raises(IndexError):
        cli.run_command_from_line("call_consensus -v 0 -f -l %s/snplist.txt -e %s/excl.vcf -o %s/one.fasta %s/reads.all.pileup" % (tmp_path, sdir, sdir, sdir))
    assert po.call_consensus_sites(b"".join(a), [(b"c1", 10), (b"c1", 120), (b"c1", 150)], set(), po.CallerParams())[0] == b"AAA"
    with pytest.raises(IndexError):
        po.call_consensus_sites(b"".join(c), [(b"c1", 10), (b"c1", 120), (b"c1", 150)], set(), po.CallerParams())


def test_vcf_all_pos_ends_at_the_first_malformed_line_of_any_kind(tmp_path, monkeypatch):
    """With --vcfAllPos the reference builds a Record from EVERY line (pileup.py:418-421): a blank or one-field line ends with
    IndexError there (with a position set the reader fails to unpack two fields: ValueError), and of several malformed lines the
    first in the file decides — a line without qualities before a line whose position is no number is IndexError, not ValueError
    (both found by tools/fuzz_campaign.py's "allpos" kind)."""
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    monkeypatch.delenv("errorOutputFile", raising=False)
    good = b"".join(b"c1\t%d\tA\t3\t...\tIII\n" % k for k in range(1, 40))
    tail = b"".join(b"c1\t%d\tA\t3\t...\tIII\n" % k for k in range(50, 70))
    sdir = _write_sample(tmp_path, "s", good)
    (tmp_path / "snplist.txt").write_text("c1\t5\t1\ts\nc1\t60\t1\ts\n")
    line = "call_consensus -v 0 -f -l %s/snplist.txt -o %s/consensus.fasta --vcfFileName all.vcf%%s %s/reads.all.pileup" % (tmp_path, sdir, sdir)
    for body, all_pos_exc, listed_exc in ((good + b"\n" + tail, IndexError, ValueError),                      # a blank line
                                          (good + b"onefield\n" + tail, IndexError, ValueError),
                                          (good + b"c1\t45\tA\t3\t..\n" + b"c1\tx\tA\t3\t...\tIII\n" + tail, IndexError, ValueError),   # (45 is not listed)
                                          (good + b"c1\tx\tA\t3\t...\tIII\n" + b"c1\t45\tA\t3\t..\n" + tail, ValueError, ValueError)):
        (sdir / "reads.all.pileup").write_bytes(body)
        with pytest.raises(all_pos_exc):
            cli.run_command_from_line(line % " --vcfAllPos")
        with pytest.raises(listed_exc):
            cli.run_command_from_line(line % "")

"""Phase-1 site calling: the device pass + host finish (snp_pipeline_amd/varscan.py) against the CPU restatement
of `VarScan mpileup2snp` (oracle/varscan_oracle.py) on seeded pileups, through the C ABI and through the console script.

What the reference's own data pins (the text and arithmetic of all 69 019 bundled var.flt.vcf lines) is checked on the CPU in
tests/test_oracle.py and tests/test_host_cpu.py; no pileup accompanies those files, so here the oracle is the checker."""
import os
import time

import numpy as np
import pytest

from oracle import fuzz
from oracle import varscan_oracle as vo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def d():
    from tests.gpu_util import get_device
    return get_device()


def _vcf(d, path, out, extra):
    from snp_pipeline_amd import varscan
    return varscan.mpileup2snp(d, path, out, varscan.Options(extra))


CASES = [("--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5", dict(vo.PIPELINE_DEFAULTS)),
         ("", {}),
         ("--min-var-freq 0.5 --min-reads2 3 --p-value 1e-6 --strand-filter 0 --min-coverage 10 --output-vcf 1",
          dict(min_var_freq=0.5, min_reads2=3, p_value=1e-6, strand_filter=0, min_coverage=10)),
         ("--min-avg-qual 0 --min-var-freq 0.3 --min-freq-for-hom 0.95", dict(min_avg_qual=0, min_var_freq=0.3, min_freq_for_hom=0.95))]


@pytest.mark.parametrize("seed,eol", [(1, b"\n"), (2, b"\r\n"), (3, b"\n")])
def test_var_flt_vcf_equals_oracle(d, tmp_path, seed, eol):
    data = fuzz.varscan_pileup(seed, 7000, eol=eol)
    path = str(tmp_path / "reads.all.pileup")
    with open(path, "wb") as f:
        f.write(data)
    for extra, kw in CASES:
        out = str(tmp_path / "var.flt.vcf")
        n_lines, n_rows = _vcf(d, path, out, extra)
        want = vo.mpileup2snp(data, vo.Params(**kw))
        got = open(out).read()
        assert got == want, (seed, extra)
        assert n_rows == sum(1 for ln in want.splitlines() if not ln.startswith("#")) and n_rows > 20
        assert n_lines == len([ln for ln in data.split(eol) if ln])


def test_deep_and_mixed_lines_take_the_other_code_paths(d, tmp_path):
    """Blocks whose lines do not fit the LDS span are walked in global memory (depth 1500-4000), blocks of 128 and 64 lines
    (mean line length), and a file that mixes 4 KiB lines with short ones: all equal to the restatement."""
    for seed, n, depths in ((21, 600, (1500, 2500, 4000, 30, 0)), (22, 3000, (150, 200, 120, 0)), (23, 3000, (300, 420, 8, 350)),
                            (24, 2500, (30, 30, 30, 30, 30, 30, 30, 3000))):
        data = fuzz.varscan_pileup(seed, n, depths=depths)
        path = str(tmp_path / "deep.pileup")
        with open(path, "wb") as f:
            f.write(data)
        for extra, kw in CASES[:2]:
            out = str(tmp_path / "deep.vcf")
            _vcf(d, path, out, extra)
            assert open(out).read() == vo.mpileup2snp(data, vo.Params(**kw)), (seed, extra)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_arbitrary_read_base_text_walks_like_the_restatement(d, tmp_path, seed):
    """Adversarial columns (fuzz.varscan_adversarial): the LDS walk, the global-memory walk and the restatement agree."""
    data = fuzz.varscan_adversarial(seed, 3000)
    if seed == 4:                                               # the same lines between very long ones: blocks that walk global memory
        long_line = b"cL\t1\tA\t9000\t" + b".,Aa" * 9000 + b"\t" + b"I5" * 18000 + b"\n"
        lines = data.split(b"\n")
        data = b"\n".join(lines[:500]) + b"\n" + long_line * 3 + b"\n".join(lines[500:])
    path = str(tmp_path / "adv.pileup")
    with open(path, "wb") as f:
        f.write(data)
    for extra, kw in CASES + [("--min-avg-qual 0 --min-var-freq 0.05 --min-reads2 1 --min-coverage 1 --strand-filter 0",
                               dict(min_avg_qual=0, min_var_freq=0.05, min_reads2=1, min_coverage=1, strand_filter=0)),
                              ("--min-avg-qual 100 --min-coverage 1 --min-reads2 1 --min-var-freq 0.01", dict(min_avg_qual=100, min_coverage=1, min_reads2=1, min_var_freq=0.01))]:
        out = str(tmp_path / "adv.vcf")
        _vcf(d, path, out, extra)
        assert open(out, encoding="latin-1").read() == vo.mpileup2snp(data, vo.Params(**kw)), (seed, extra)


def test_many_files_in_one_call_equal_single_calls(d, tmp_path):
    """snpgpu_varscan_files: files of very different sizes, an empty one, a missing one, a malformed one, one with more
    records than the shared array holds — each result equals the single-file call's."""
    from snp_pipeline_amd import varscan
    from snp_pipeline_amd.device import PileupFormatError, PileupIOError
    opts = varscan.Options("--min-var-freq 0.2 --min-reads2 2")
    blobs = [fuzz.varscan_pileup(31, 9000), fuzz.varscan_pileup(32, 300), b"", fuzz.varscan_pileup(33, 40000, depths=(20, 30, 30, 45)),
             b"c\t1\tA\t9\tGGGGGGGGG\n", fuzz.varscan_pileup(34, 2500, eol=b"\r\n"), fuzz.varscan_pileup(35, 1)]
    paths = []
    for i, data in enumerate(blobs):
        path = str(tmp_path / ("f%d.pileup" % i))
        with open(path, "wb") as f:
            f.write(data)
        paths.append(path)
    paths.insert(3, str(tmp_path / "missing.pileup"))
    for capacity in (32768, 64):                                 # 64: most files overflow the shared array and are repeated alone
        got = d.varscan_files(paths, opts.device_params(), capacity=capacity)
        assert len(got) == len(paths)
        for path, (recs, n_lines) in zip(paths, got):
            if path.endswith("missing.pileup"):
                assert isinstance(recs, PileupIOError)
            elif path.endswith("f4.pileup"):
                assert isinstance(recs, PileupFormatError) and "byte 0 " in str(recs)
            else:
                want, want_lines = d.varscan_file(path, opts.device_params())
                assert n_lines == want_lines and recs.tobytes() == want.tobytes(), path
    vcfs = [p + ".vcf" for p in paths]
    res = varscan.mpileup2snp_files(d, paths, vcfs, opts)
    for path, vcf, r, data in zip([p for p in paths if "missing" not in p], [v for v in vcfs if "missing" not in v],
                                  [r for p, r in zip(paths, res) if "missing" not in p], blobs):
        if isinstance(r, Exception):
            assert path.endswith("f4.pileup")
            continue
        assert open(vcf, "rb").read().decode("latin-1") == vo.mpileup2snp(data, vo.Params(min_var_freq=0.2, min_reads2=2)), path


def test_records_capacity_retry_and_order(d, tmp_path):
    from snp_pipeline_amd import varscan
    data = fuzz.varscan_pileup(5, 5000)
    path = str(tmp_path / "p.pileup")
    with open(path, "wb") as f:
        f.write(data)
    prm = varscan.Options("--min-var-freq 0.01 --min-reads2 1").device_params()
    a, n_lines = d.varscan_file(path, prm, capacity=1)                 # grows until everything fits
    b, _ = d.varscan_file(path, prm, capacity=1 << 16)
    assert len(a) == len(b) > 300 and a.tobytes() == b.tobytes()
    key = a["line_off"].astype(np.int64) * 256 + a["alt_base"]
    assert (np.diff(key) > 0).all()                                    # file order, alleles A < C < G < T within a line
    assert (a["adf"] + a["adr"] > 0).all() and (a["total"] >= a["adf"] + a["adr"] + a["rdf"] + a["rdr"]).all()


def test_malformed_lines_are_refused_like_the_oracle(d, tmp_path):
    from snp_pipeline_amd import varscan
    from snp_pipeline_amd.device import PileupFormatError, PileupIOError
    good = b"c\t1\tA\t9\tGGGGGGGGG\tIIIIIIIII\n"
    for bad in (b"c\t2\tA\t9\tGGGGGGGGG\n", b"c\t2\tA\tx9\tGGGGGGGGG\tIIIIIIIII\n", b"c 2 A 9 GGGGGGGGG IIIIIIIII\n", b"c\t2\tAC\t9\tGGGGGGGGG\tIIIIIIIII\n",
                b"c\t2\tA\t9\tGGGGGGGGG\t\n", b"\t2\tA\t9\tGGGGGGGGG\tIIIIIIIII\n"):
        path = str(tmp_path / "bad.pileup")
        data = good * 3 + bad + good
        with open(path, "wb") as f:
            f.write(data)
        with pytest.raises(PileupFormatError) as e:
            _vcf(d, path, str(tmp_path / "o.vcf"), "")
        assert "byte %d " % (3 * len(good)) in str(e.value)
        with pytest.raises(ValueError):
            vo.mpileup2snp(data, vo.Params())
    with pytest.raises(PileupIOError):
        _vcf(d, str(tmp_path / "missing.pileup"), str(tmp_path / "o.vcf"), "")
    # an empty file and a file of empty lines give the bare header
    for data in (b"", b"\n\n\n"):
        path = str(tmp_path / "e.pileup")
        with open(path, "wb") as f:
            f.write(data)
        assert _vcf(d, path, str(tmp_path / "o.vcf"), "")[1] == 0
        assert open(str(tmp_path / "o.vcf")).read() == vo.VCF_HEADER % {"q": 15} == varscan.header_text(15)


def test_call_sites_console_script(tmp_path, monkeypatch):
    """cfsan_snp_pipeline call_sites with a fresh pileup (samtools is not run, call_sites.py:70-72): var.flt.vcf from the device,
    VarscanMpileup2snp_ExtraParams honoured, the freshness check on the second run, sample errors for a missing BAM."""
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    ref = tmp_path / "ref.fasta"
    ref.write_text(">ctgA\nACGT\n")
    sdir = tmp_path / "samples" / "s1"
    sdir.mkdir(parents=True)
    bam = sdir / "reads.sorted.deduped.indelrealigned.bam"
    bam.write_bytes(b"not really a bam")
    old = time.time() - 100
    os.utime(str(bam), (old, old))
    os.utime(str(ref), (old, old))
    data = fuzz.varscan_pileup(11, 4000)
    (sdir / "reads.all.pileup").write_bytes(data)
    monkeypatch.setenv("VarscanMpileup2snp_ExtraParams", "--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5")
    monkeypatch.chdir(tmp_path)
    cli.run_command_from_args(cli.parse_command_line("call_sites -v 0 %s %s" % (ref, sdir)))
    want = vo.mpileup2snp(data, vo.Params(**vo.PIPELINE_DEFAULTS))
    vcf = sdir / "var.flt.vcf"
    assert vcf.read_text() == want
    # fresh: not rebuilt
    stamp = os.stat(str(vcf)).st_mtime_ns
    monkeypatch.setenv("VarscanMpileup2snp_ExtraParams", "--min-var-freq 0.5")
    cli.run_command_from_args(cli.parse_command_line("call_sites -v 0 %s %s" % (ref, sdir)))
    assert os.stat(str(vcf)).st_mtime_ns == stamp
    os.utime(str(vcf), (old + 50, old + 50))                             # now older than the pileup: rebuilt (-f would also re-run samtools)
    cli.run_command_from_args(cli.parse_command_line("call_sites -v 0 %s %s" % (ref, sdir)))
    assert vcf.read_text() == vo.mpileup2snp(data, vo.Params(min_var_freq=0.5))
    # the result feeds the next stage's reader
    from snp_pipeline_amd import utils
    assert len(utils.convert_vcf_file_to_snp_set(str(vcf))) == sum(1 for ln in vcf.read_text().splitlines() if not ln.startswith("#"))


def test_full_size_sample_runs_and_matches_oracle_on_its_variant_lines(d, tmp_path):
    """One 5 Mbp x 30x sample (432 MB) written to disk: every record's line re-parsed by the oracle gives the same row, and
    every planted homozygous site with enough depth is found."""
    import torch
    from snp_pipeline_amd import varscan
    G, S = 5_000_000, 5_000
    d.use_torch_stream()
    ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda")
    d.synth_reference_dev(1, G, ref.data_ptr())
    rng = np.random.default_rng(4)
    pos = np.sort(rng.choice(np.arange(501, G - 499), size=S, replace=False))
    refh = ref.cpu().numpy()
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    alt_h = np.zeros(G + 1, dtype=np.uint8)
    alt_h[pos] = acgt[(np.searchsorted(acgt, refh[pos]) + 1 + rng.integers(0, 3, size=S)) % 4]
    alt = torch.from_numpy(alt_h).cuda()
    n = d.synth_pileup_dev(3, 0, G, ref.data_ptr(), alt.data_ptr(), 0, 0, p_same=1.0, p_other=1.0)
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    assert d.synth_pileup_dev(3, 0, G, ref.data_ptr(), alt.data_ptr(), buf.data_ptr(), n + 64, p_same=1.0, p_other=1.0) == n
    data = buf[:n].cpu().numpy().tobytes()
    path = str(tmp_path / "reads.all.pileup")
    with open(path, "wb") as f:
        f.write(data)
    out = str(tmp_path / "var.flt.vcf")
    t0 = time.time()
    n_lines, n_rows = _vcf(d, path, out, "--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5")
    wall = time.time() - t0
    assert n_lines == data.count(b"\n")
    rows = [ln for ln in open(out).read().splitlines(True) if not ln.startswith("#")]
    assert len(rows) == n_rows and n_rows > 0.5 * S
    prm = vo.Params(**vo.PIPELINE_DEFAULTS)
    recs, _ = d.varscan_file(path, varscan.Options("--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5").device_params())
    assert len(recs) == n_rows
    for k in rng.choice(n_rows, size=400, replace=False):
        off = int(recs["line_off"][k])
        f = data[off:data.index(b"\n", off)].split(b"\t")
        r = vo.call_line(f[2].decode(), int(f[3]), f[4], f[5], prm)
        assert r is not None and vo.vcf_row(f[0].decode(), f[1].decode(), r) == rows[k]
    # lines that are NOT in the output: the oracle calls nothing on a sample of them either
    called = set(int(x) for x in recs["line_off"])
    starts = [0] + [i + 1 for i in rng.choice(n - 400, size=300)]
    for s in starts:
        s = data.index(b"\n", s) + 1 if s else 0
        if s in called:
            continue
        f = data[s:data.index(b"\n", s)].split(b"\t")
        assert vo.call_line(f[2].decode(), int(f[3]), f[4], f[5], prm) is None
    print("full-size sample: %d lines, %d sites, %.2f s wall (%.1f GB/s file -> var.flt.vcf)" % (n_lines, n_rows, wall, n / wall / 1e9))


def test_call_sites_batch_equals_per_sample_cli(tmp_path, monkeypatch):
    """call_sites_batch (extension): every sample of a sampleDirsFile in one process = the per-sample subcommand; a sample
    without a BAM and one with a malformed pileup are sample errors, the others still get their var.flt.vcf."""
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    ref = tmp_path / "ref.fasta"
    ref.write_text(">ctgA\nACGT\n")
    old = time.time() - 100
    os.utime(str(ref), (old, old))
    dirs, blobs = [], []
    for i in range(5):
        sdir = tmp_path / "samples" / ("s%d" % i)
        sdir.mkdir(parents=True)
        if i != 3:
            bam = sdir / "reads.sorted.deduped.indelrealigned.bam"
            bam.write_bytes(b"placeholder")
            os.utime(str(bam), (old, old))
        data = fuzz.varscan_pileup(50 + i, 3000 + 500 * i) if i != 4 else b"c\t1\tA\t9\tGGGGGGGGG\n"
        (sdir / "reads.all.pileup").write_bytes(data)
        dirs.append(str(sdir))
        blobs.append(data)
    dirs_file = tmp_path / "sampleDirectories.txt"
    dirs_file.write_text("\n".join(dirs) + "\n")
    monkeypatch.setenv("VarscanMpileup2snp_ExtraParams", "--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5")
    monkeypatch.setenv("StopOnSampleError", "false")
    monkeypatch.setenv("errorOutputFile", str(tmp_path / "error.log"))
    monkeypatch.chdir(tmp_path)
    args = cli.parse_command_line("call_sites_batch -v 0 %s %s" % (ref, dirs_file))
    assert cli.run_command_from_args(args) == 0
    log = (tmp_path / "error.log").read_text()
    assert "s3/reads.sorted.deduped.indelrealigned.bam" in log and "call_sites failed for sample s4: PileupFormatError" in log
    for i in (0, 1, 2):
        want = vo.mpileup2snp(blobs[i], vo.Params(**vo.PIPELINE_DEFAULTS))
        assert open(os.path.join(dirs[i], "var.flt.vcf")).read() == want
    assert not os.path.exists(os.path.join(dirs[3], "var.flt.vcf"))
    # the per-sample subcommand finds the batch's files fresh and leaves them alone
    stamp = os.stat(os.path.join(dirs[0], "var.flt.vcf")).st_mtime_ns
    cli.run_command_from_args(cli.parse_command_line("call_sites -v 0 %s %s" % (ref, dirs[0])))
    assert os.stat(os.path.join(dirs[0], "var.flt.vcf")).st_mtime_ns == stamp


def test_a_lone_carriage_return_ends_a_line_as_for_javas_readline(d, tmp_path):
    """BufferedReader.readLine() ends a line at LF, CR or CR LF.  A CR in the middle of the read bases therefore makes two lines,
    both with too few columns: VarScan stops with "Invalid format for pileup", the device pass and the restatement refuse the file
    alike; a CR between two complete lines is a line end like any other (found by tools/fuzz_campaign.py: the restatement split
    at LF only)."""
    from snp_pipeline_amd import varscan
    from snp_pipeline_amd.device import PileupFormatError
    data = fuzz.varscan_pileup(77, 400)
    lines = data.split(b"\n")
    k = next(i for i, ln in enumerate(lines) if ln.count(b"\t") == 5 and len(ln.split(b"\t")[4]) > 12)
    f = lines[k].split(b"\t")
    f[4] = f[4][:6] + b"\r" + f[4][6:]
    broken = b"\n".join(lines[:k] + [b"\t".join(f)] + lines[k + 1:])
    path = str(tmp_path / "p.pileup")
    with open(path, "wb") as fh:
        fh.write(broken)
    with pytest.raises(ValueError):
        vo.mpileup2snp(broken, vo.Params(**vo.PIPELINE_DEFAULTS))
    with pytest.raises(PileupFormatError):
        varscan.mpileup2snp(d, path, str(tmp_path / "p.vcf"), varscan.Options(CASES[0][0]))
    mixed = b"".join(ln + (b"\r" if i % 3 == 0 else b"\r\n" if i % 3 == 1 else b"\n") for i, ln in enumerate(lines) if ln)
    with open(path, "wb") as fh:
        fh.write(mixed)
    varscan.mpileup2snp(d, path, str(tmp_path / "p.vcf"), varscan.Options(CASES[0][0]))
    assert open(str(tmp_path / "p.vcf")).read() == vo.mpileup2snp(mixed, vo.Params(**CASES[0][1])) == vo.mpileup2snp(data, vo.Params(**CASES[0][1]))


def test_empty_read_base_and_quality_columns_are_columns(d, tmp_path):
    """String.split("\\t") drops TRAILING empty strings only: a line with an empty read-base column, or an empty quality column
    followed by a seventh column, still has more than five columns and is processed (it can call nothing); an empty quality
    column at the end of the line leaves five and is "Invalid format" (found by tools/fuzz_campaign.py: the kernels asked for six
    non-empty columns)."""
    from snp_pipeline_amd import varscan
    from snp_pipeline_amd.device import PileupFormatError
    data = fuzz.varscan_pileup(78, 300)
    lines = [ln for ln in data.split(b"\n") if ln]
    extra = [b"ctgA\t9001\tA\t3\t\tGGG,,,\tIIIIII", b"ctgA\t9002\tA\t12\tGGGGGGGGGGGG\t\tIIIIIIIIIIII", b"ctgA\t9003\tA\t12\t\t\tx"]
    ok = b"\n".join(lines[:100] + extra + lines[100:]) + b"\n"
    path = str(tmp_path / "p.pileup")
    opts = "--min-avg-qual 0 --min-var-freq 0.05 --min-reads2 1 --min-coverage 1 --strand-filter 0"
    kw = dict(min_avg_qual=0, min_var_freq=0.05, min_reads2=1, min_coverage=1, strand_filter=0)
    with open(path, "wb") as fh:
        fh.write(ok)
    varscan.mpileup2snp(d, path, str(tmp_path / "p.vcf"), varscan.Options(opts))
    assert open(str(tmp_path / "p.vcf")).read() == vo.mpileup2snp(ok, vo.Params(**kw))
    for bad_line in (b"ctgA\t9004\tA\t12\tGGGGGGGGGGGG\t", b"ctgA\t9004\tA\t12\tGGGGGGGGGGGG\t\t\t", b"ctgA\t9004\tA\t12\t\t"):
        bad = b"\n".join(lines[:100] + [bad_line] + lines[100:]) + b"\n"
        with open(path, "wb") as fh:
            fh.write(bad)
        with pytest.raises(ValueError):
            vo.mpileup2snp(bad, vo.Params(**kw))
        with pytest.raises(PileupFormatError):
            varscan.mpileup2snp(d, path, str(tmp_path / "p.vcf"), varscan.Options(opts))


def test_full_candidate_list_unaligned_text_and_lines_across_tiles(d, tmp_path):
    """The index-free pass (k_varscan_scan): (1) more candidate lines than the list holds — depth-0 lines, which the
    shortcut cannot vouch for, by the ten thousand — are looked at on the spot and the answer is the same; (2) text at every
    alignment mod 16 in device memory (snpgpu_varscan_dev); (3) line terminators, TABs and whole lines placed on the edges of the
    4 KiB tiles and of the ring of three tile slots, with LF / CR LF / lone CR ends."""
    import torch
    from snp_pipeline_amd import varscan
    opts = varscan.Options("--min-var-freq 0.2 --min-reads2 2")
    prm = opts.device_params()
    kw = dict(min_var_freq=0.2, min_reads2=2)
    # (1)
    rng = np.random.default_rng(5)
    real = fuzz.varscan_pileup(41, 3000).split(b"\n")
    lines = []
    for i in range(70000):
        lines.append(b"c0\t%d\tA\t0\t*\t*" % (i + 1))
        if i % 23 == 0:
            lines.append(real[(i // 23) % len(real)])
    data = b"\n".join(ln for ln in lines if ln) + b"\n"
    path = str(tmp_path / "many.pileup")
    with open(path, "wb") as f:
        f.write(data)
    out = str(tmp_path / "many.vcf")
    n_lines, _ = varscan.mpileup2snp(d, path, out, opts)
    assert open(out).read() == vo.mpileup2snp(data, vo.Params(**kw)) and n_lines == data.count(b"\n")
    # (2)
    data = fuzz.varscan_pileup(42, 4000)
    want, want_lines = None, None
    buf = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
    for shift in range(16):
        buf.zero_()
        buf[shift:shift + len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        recs, nl = d.varscan_dev(buf.data_ptr() + shift, len(data), prm)
        if want is None:
            want, want_lines = recs.tobytes(), nl
            assert nl == len([ln for ln in data.split(b"\n") if ln]) and len(recs) > 50
        assert recs.tobytes() == want and nl == want_lines, shift
    # (3) a variant line pushed byte by byte across a tile edge, three line ends, with padding lines of chosen length in front
    variant = b"c1\t7\tA\t12\tGGGGGGgggggg\tIIIIIIIIIIII"
    plainl = b"c1\t6\tA\t12\t......,,,,,,\tIIIIIIIIIIII"
    for eol in (b"\n", b"\r\n", b"\r"):
        for edge in (4096, 8192, 12288, 16384):                  # (12288: where a wave's ring of three tile slots wraps around)
            for delta in range(-len(variant) - 3, 4):
                pad_total = edge + delta
                head = []
                used = 0
                while pad_total - used > 2 * (len(plainl) + len(eol)):
                    head.append(plainl)
                    used += len(plainl) + len(eol)
                rest = pad_total - used - len(eol)
                if rest >= 22:                                   # a last padding line of exactly the missing length
                    head.append(b"c1\t5\tA\t%d\t%s\t%s" % ((rest - 10) // 2, b"." * ((rest - 10) // 2), b"I" * ((rest - 10) - (rest - 10) // 2)))
                text = eol.join(head + [variant, plainl, variant]) + eol
                p2 = str(tmp_path / "edge.pileup")
                with open(p2, "wb") as f:
                    f.write(text)
                o2 = str(tmp_path / "edge.vcf")
                try:
                    varscan.mpileup2snp(d, p2, o2, opts)
                    got = open(o2).read()
                except Exception as e:                           # noqa: B902 — compared by class below
                    got = type(e).__name__
                try:
                    want2 = vo.mpileup2snp(text, vo.Params(**kw))
                except Exception as e:                           # noqa: B902
                    want2 = type(e).__name__
                if want2 in ("ValueError", "IndexError"):
                    assert got == "PileupFormatError", (eol, edge, delta)
                else:
                    assert got == want2, (eol, edge, delta)


@pytest.mark.parametrize("eol", [b"\r", b"\n", b"\r\n"])
def test_bytes_above_0x89_next_to_terminators_and_tabs(d, tmp_path, eol):
    """The SWAR terminator test of k_varscan_scan must not carry from one byte into the next: a quality column that ends with a
    byte >= 0x8A right in front of the line terminator (lone CR: the fast form would have missed the line end), and a contig name
    that ends with one right in front of the first TAB (the TAB would have looked like a terminator)."""
    from snp_pipeline_amd import varscan
    opts = varscan.Options("--min-var-freq 0.2 --min-reads2 2 --min-avg-qual 0")
    lines = []
    for i in range(4000):
        depth = 10 + i % 7
        bases = ("G" * depth if i % 9 == 0 else "." * depth).encode()
        quals = bytes([0x49] * (depth - 1) + [0xFE if i % 2 else 0x8A])
        lines.append(b"ctg\xe9\t%d\tA\t%d\t%s\t%s" % (i + 1, depth, bases, quals))
    data = eol.join(lines) + eol
    path = str(tmp_path / "hi.pileup")
    with open(path, "wb") as f:
        f.write(data)
    out = str(tmp_path / "hi.vcf")
    n_lines, n_rows = varscan.mpileup2snp(d, path, out, opts)
    want = vo.mpileup2snp(data, vo.Params(min_var_freq=0.2, min_reads2=2, min_avg_qual=0))
    assert open(out, "rb").read().decode("latin-1") == want and n_lines == len(lines) and n_rows > 400


def test_many_resident_pileups_in_one_launch_equal_single_calls(d):
    """snpgpu_varscan_batch_dev: one scan launch over pileups of very different sizes, line ends and depths, an empty one, a
    malformed one, one with more records than the shared array holds, at odd alignments in device memory — every result equals
    the single-pileup call's (and through it the restatement's: the tests above)."""
    import torch
    from snp_pipeline_amd import varscan
    from snp_pipeline_amd.device import PileupFormatError
    prm = varscan.Options("--min-var-freq 0.2 --min-reads2 2").device_params()
    blobs = [fuzz.varscan_pileup(51, 9000), fuzz.varscan_pileup(52, 300), b"", fuzz.varscan_pileup(53, 30000, depths=(20, 30, 30, 45)),
             b"c\t1\tA\t9\tGGGGGGGGG\n", fuzz.varscan_pileup(54, 2500, eol=b"\r\n"), fuzz.varscan_pileup(55, 1),
             fuzz.varscan_pileup(56, 700, depths=(1500, 2500, 30, 0)), fuzz.varscan_pileup(57, 4000, eol=b"\r"), fuzz.varscan_adversarial(3, 2000)]
    total = sum(len(b) + 64 for b in blobs)
    buf = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    ptrs, sizes, at = [], [], 3
    for i, data in enumerate(blobs):
        if data:
            buf[at:at + len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        ptrs.append(buf.data_ptr() + at)
        sizes.append(len(data))
        at += len(data) + 17 + i                                   # every alignment class
    torch.cuda.synchronize()
    singles = []
    for p, n in zip(ptrs, sizes):
        try:
            singles.append(d.varscan_dev(p, n, prm))
        except PileupFormatError as e:
            singles.append(e)
    for capacity in (32768, 64):                                  # 64: most pileups overflow the shared array and are repeated alone
        got = d.varscan_batch_dev(ptrs, sizes, prm, capacity=capacity)
        assert len(got) == len(blobs)
        for i, (g, w) in enumerate(zip(got, singles)):
            if isinstance(w, Exception):
                assert isinstance(g, PileupFormatError) and str(g) == str(w), i
            else:
                assert g[1] == w[1] and g[0].tobytes() == w[0].tobytes(), i
    assert isinstance(singles[4], PileupFormatError) and singles[2][1] == 0 and len(singles[0][0]) > 50
    # the same pileup many times over: more files than a launch has workgroups to spare, every copy the same answer
    many = d.varscan_batch_dev([ptrs[1]] * 300 + [ptrs[0]] * 3, [sizes[1]] * 300 + [sizes[0]] * 3, prm)
    for g in many[:300]:
        assert g[1] == singles[1][1] and g[0].tobytes() == singles[1][0].tobytes()
    for g in many[300:]:
        assert g[0].tobytes() == singles[0][0].tobytes()

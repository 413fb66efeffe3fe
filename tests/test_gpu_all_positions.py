"""call_consensus --vcfAllPos on its way out of the device: the 32-byte line records against the full ones, and the
library's file-to-file writer (snpgpu_write_all_positions_vcf) against the row-by-row Python statement of the same layout
(vcf_writer.write_all_positions_vcf <- vcf_writer.py:381-435 of the reference), which the oracle tests pin."""
import argparse
import random

import numpy as np
import pytest

from oracle import fuzz
from oracle import pileup_oracle as po
from snp_pipeline_amd import _lib as L
from snp_pipeline_amd import device as dev
from snp_pipeline_amd import vcf_writer
from tests.gpu_util import get_device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def d():
    return get_device()


def _args(**kw):
    a = argparse.Namespace(minBaseQual=0, minConsFreq=0.6, minConsDpth=3, minConsStrdDpth=0, minConsStrdBias=0.0, vcfRefName="ref.fasta",
                           vcfPreserveRefCase=False, vcfFailedSnpGt=".")
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _params(a):
    return dev.make_params(a.minBaseQual, a.minConsFreq, a.minConsDpth, a.minConsStrdDpth, a.minConsStrdBias)


def _odd_lines(rng):
    """Lines the packed record cannot hold (more than three symbols, a reference field of several bytes, a depth outside 32 bits) and
    lines whose first two columns int() / split() take in their less usual spellings."""
    out = []
    for i in range(60):
        k = rng.choice((4, 5, 9, 12))
        syms = rng.sample("ACGTNRYKMSWBDHV*", k)
        reads = [rng.choice(syms) for _ in range(rng.randint(k, 40))] + syms
        rng.shuffle(reads)
        bases = "".join(c.lower() if c != "*" and rng.random() < 0.5 else c for c in reads)
        ref = rng.choice(("AC", "g,", "ac.")) if i % 6 == 5 else rng.choice("ACGTacgt")
        if len(ref) > 1:
            bases = bases.replace("*", ".")
        depth = "-%d" % len(reads) if i % 9 == 4 else ("%d" % (5_000_000_000 + i) if i % 9 == 7 else "%d" % len(reads))
        out.append("%s\t%s\t%s\t%s" % (ref, depth, bases, "".join(chr(33 + rng.randint(0, 40)) for _ in reads)))
    return out


def _mixed_pileup(seed, genome_len, odd_every=97):
    """A fuzzed well-formed pileup with the odd lines of _odd_lines spliced in; positions written as 007 / +12 / 1_0 on some lines."""
    rng = random.Random(seed)
    data, _, sites = fuzz.synth_pileup(seed, genome_len=genome_len, n_sites=max(20, genome_len // 50))
    odd = _odd_lines(rng)
    lines = data.split(b"\n")[:-1]
    out = []
    for i, ln in enumerate(lines):
        f = ln.split(b"\t")
        if i % odd_every == odd_every - 1:
            f = f[:2] + odd[(i // odd_every) % len(odd)].encode().split(b"\t")
        if i % 53 == 7:
            f[1] = b"00" + f[1]
        elif i % 53 == 19:
            f[1] = b"+" + f[1]
        elif i % 53 == 31 and len(f[1]) > 1:
            f[1] = f[1][:1] + b"_" + f[1][1:]
        out.append(b"\t".join(f))
    return b"\n".join(out) + b"\n", sites


def _same_records(a, spill_a, b, spill_b):
    """Two calls' records of the same lines: equal byte for byte but for the INDEX of a line's spill record (bits 8-31 of n_symbols:
    the records of a call are claimed with an atomic, in whatever order its waves get there) — what the index points at is compared."""
    plain_a, plain_b = a.copy(), b.copy()
    plain_a["n_symbols"] &= 0xFF
    plain_b["n_symbols"] &= 0xFF
    assert plain_a.tobytes() == plain_b.tobytes()
    code_a, code_b = a["n_symbols"] >> 8, b["n_symbols"] >> 8
    assert np.array_equal(code_a != 0, code_b != 0)
    size = dev.SPILL_DTYPE.itemsize
    for i in np.nonzero(code_a)[0]:
        ra, rb = spill_a[int(code_a[i]) - 1], spill_b[int(code_b[i]) - 1]
        n, ref_len = int(ra["n"]), int(ra["ref_len"])
        assert (n, ref_len, int(ra["depth64"])) == (int(rb["n"]), int(rb["ref_len"]), int(rb["depth64"])), i
        for name in ("sym", "total", "fwd", "rev"):                  # (what lies past the entries in use is whatever the arena held before)
            assert np.array_equal(ra[name][:n], rb[name][:n]), (i, name)
        assert ra["ref"][:min(ref_len, L.SPILL_REF)].tobytes() == rb["ref"][:min(ref_len, L.SPILL_REF)].tobytes(), i
        more = (ref_len - L.SPILL_REF + size - 1) // size if ref_len > L.SPILL_REF else 0
        for k in range(1, more + 1):                                 # a reference field longer than one record goes on in the next ones
            assert spill_a[int(code_a[i]) - 1 + k].tobytes() == spill_b[int(code_b[i]) - 1 + k].tobytes()
    assert int((code_a != 0).sum()) > 0


@pytest.mark.parametrize("seed, genome_len", [(1, 3000), (2, 70000), (3, 140000)])
def test_line_records_of_32_bytes_equal_the_full_ones(d, tmp_path, seed, genome_len):
    data, sites = _mixed_pileup(seed, genome_len)
    path = str(tmp_path / "reads.all.pileup")
    with open(path, "wb") as f:
        f.write(data)
    ss = d.siteset(sites, [L.SITE_IN_SNPLIST | (L.SITE_EXCLUDED if i % 5 == 0 else 0) for i in range(len(sites))])
    prm = _params(_args())
    off, flags, counts = d.call_all_lines(ss, path, prm, check=False)
    spill = d.last_spill
    off2, recs, widx, wide = d.call_all_lines_compact(ss, path, prm, capacity=len(off) // 2, wide_capacity=1)      # (both arrays too small at first)
    spill2 = d.last_spill
    assert len(off) == data.count(b"\n") and np.array_equal(off, off2)
    flags2, counts2 = dev.expand_line_records(recs, widx, wide)
    assert np.array_equal(flags, flags2)
    _same_records(counts, spill, counts2, spill2)
    # nearly every line is packed; the odd ones are wide, in file order
    n_wide = len(widx)
    assert 0 < n_wide < len(off) // 20 and np.all(np.diff(widx.astype(np.int64)) > 0)
    assert n_wide >= len(off) // 97 - 1
    assert (recs["n_symbols"][recs["n_symbols"] != dev.LINE_WIDE] <= dev.LINE_SYMS).all()
    # the device's packing (k_compact_lines, k_gather_wide) is device.pack_line_records', record for record
    recs_np, widx_np, _ = dev.pack_line_records(counts, flags)
    assert np.array_equal(widx, widx_np) and recs.tobytes() == recs_np.tobytes()


@pytest.mark.parametrize("seed, genome_len, kw", [
    (4, 3000, {}),
    (5, 70000, {"vcfPreserveRefCase": True, "vcfFailedSnpGt": "1", "minBaseQual": 20, "minConsStrdDpth": 2, "minConsStrdBias": 0.1}),
    (6, 200000, {"vcfFailedSnpGt": "0", "minConsDpth": 25}),           # (more lines than three pieces of the read-back hold)
])
def test_file_to_file_writer_equals_the_row_by_row_one(d, tmp_path, seed, genome_len, kw):
    data, sites = _mixed_pileup(seed, genome_len)
    if seed == 4:                                                    # CR LF ends, a contig name of 700 bytes, leading blanks before CHROM
        long_name = b"ctg_" + b"x" * 700
        data = data.replace(b"\n", b"\r\n") + long_name + b"\t17\tA\t3\t.,.\tIII\r\n" + b"  synth_chr1\t99999\tC\t2\t..\tII\r\n"
    path = str(tmp_path / "reads.all.pileup")
    with open(path, "wb") as f:
        f.write(data)
    args = _args(**kw)
    prm = _params(args)
    ss = d.siteset(sites, [L.SITE_IN_SNPLIST | (L.SITE_EXCLUDED if i % 5 == 0 else 0) for i in range(len(sites))])
    off, flags, counts = d.call_all_lines(ss, path, prm, check=True)
    spill = d.last_spill
    for only_listed in (False, True):
        keep = np.nonzero(flags)[0] if only_listed else np.arange(len(off))
        want_path, got_path = str(tmp_path / "want.vcf"), str(tmp_path / "got.vcf")
        vcf_writer.write_all_positions_vcf(want_path, "sampleA", args, path, off[keep], counts[keep], spill=spill)
        n_lines, n_rows = vcf_writer.write_all_positions_vcf_from_pileup(d, ss, got_path, "sampleA", args, path, prm, only_listed=only_listed, check=True)
        assert (n_lines, n_rows) == (len(off), len(keep))
        want, got = open(want_path, "rb").read(), open(got_path, "rb").read()
        assert len(got) == len(want)
        if got != want:
            w, g = want.split(b"\n"), got.split(b"\n")
            k = next(i for i in range(len(w)) if w[i] != g[i])
            raise AssertionError("row %d differs:\n%r\n%r" % (k, w[k], g[k]))
    assert n_rows < n_lines                                           # (the listed lines are a part of the file)


def test_the_first_line_the_reference_cannot_take_ends_the_file_to_file_writer_too(d, tmp_path):
    """--vcfAllPos builds a Record from EVERY line (pileup.py:418-421): a line with three fields raises IndexError, a depth of 'x7'
    ValueError, whichever comes first in the file — also when a malformed position column follows further down; nothing is written."""
    good = b"c1\t%d\tA\t3\t.,.\tIII\n"
    body = b"".join(good % i for i in range(1, 400))
    args = _args()
    prm = _params(args)
    ss = d.siteset([(b"c1", 7)], [L.SITE_IN_SNPLIST])
    cases = [
        (body + b"c1\t400\tA\n" + body, IndexError),                                          # three fields
        (body + b"c1\t400\tA\tx7\t.\tI\n" + b"c1\t401\tA\n", ValueError),                     # the earlier of two Record-level failures
        (body + b"c1\t400\tA\t2\t..\n" + body, IndexError),                                   # depth > 0 and no quality column
        (body + b"c1\t400\tA\n" + body + b"c1\t12x\tA\t1\t.\tI\n", IndexError),              # a bad position column later in the file
        (body + b"c1\t12x\tA\t1\t.\tI\n" + body + b"c1\t400\tA\n", ValueError),              # ... and earlier
    ]
    for k, (data, exc) in enumerate(cases):
        path, out = str(tmp_path / ("p%d.pileup" % k)), str(tmp_path / ("o%d.vcf" % k))
        with open(path, "wb") as f:
            f.write(data)
        with pytest.raises(dev.PileupFormatError) as info:
            vcf_writer.write_all_positions_vcf_from_pileup(d, ss, out, "s", args, path, prm, check=True)
        assert info.value.reference_exception is exc, (k, info.value)
        import os
        assert not os.path.exists(out)
    # without the check (the repeated-positions case looks only at its own lines) the rows of the well-formed lines are written
    path, out = str(tmp_path / "p0.pileup"), str(tmp_path / "unchecked.vcf")
    n_lines, n_rows = vcf_writer.write_all_positions_vcf_from_pileup(d, ss, out, "s", args, path, prm, only_listed=True, check=False)
    assert (n_lines, n_rows) == (2 * 399 + 1, 2)
    # an output that cannot be written ends as open(path, "w") of the reference's writer does (vcf_writer.py:92-99): an OSError with the path
    with pytest.raises(OSError) as info:
        vcf_writer.write_all_positions_vcf_from_pileup(d, ss, str(tmp_path / "no_such_dir" / "x.vcf"), "s", args, path, prm, only_listed=True, check=False)
    assert "no_such_dir" in str(info.value)
    # an empty pileup: the header alone
    empty = str(tmp_path / "empty.pileup")
    open(empty, "wb").close()
    assert vcf_writer.write_all_positions_vcf_from_pileup(d, ss, out, "s", args, empty, prm) == (0, 0)
    assert open(out).read().splitlines()[-1].startswith("#CHROM")

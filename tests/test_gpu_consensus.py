"""HIP pileup scan + consensus caller (through the C ABI) against the golden vectors and the oracle."""
import random

import numpy as np
import pytest

from oracle import fuzz
from oracle import pileup_oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def d():
    from tests.gpu_util import get_device
    return get_device()


def _names(mask, p):
    names = po.filter_names(p)
    return [names[i] for i in range(6) if mask >> i & 1] or None


def test_golden_record_vectors(d, pileup_vectors):
    """Every fuzzed line the real reference parsed: counts, ranking, consensus and filter list."""
    checked, spilled, long_refs = _golden_records(d, pileup_vectors["records"])
    assert checked > 15000 and spilled > 100 and long_refs == 0


def test_golden_reference_fields_of_several_bytes(d, longref_vectors):
    """The same for lines whose reference-base field has several bytes (the real reference's Records): every '.' / ','
    counts once per character of the field, the field itself travels in the position's spill record."""
    checked, spilled, long_refs = _golden_records(d, longref_vectors["records"])
    assert checked > 1500 and long_refs == checked
    # depth columns outside 0 .. 2^32 - 1 (closed in round 4: the value rides in the spill record's depth64)
    checked, _, _ = _golden_records(d, longref_vectors["wide_depth_records"])
    assert checked > 60 and _golden_records.wide_depths >= 15
    # fields of 65 to 20 000 bytes (refused up to round 4): they go on in the spill records behind the position's own
    very = longref_vectors["very_long_ref_records"]
    checked, _, long_refs = _golden_records(d, very)
    assert checked > 100 and long_refs == checked


def _golden_records(d, recs):
    from snp_pipeline_amd import _lib as L
    from tests.gpu_util import gpu_consensus
    param_sets = sorted({tuple(c["params"]) for v in recs for q in v["by_q"].values() if "calls" in q for c in q["calls"]})
    checked = spilled = long_refs = wide = 0
    for params in param_sets:
        p = po.CallerParams(*params)
        q = str(params[0])
        good = [v for v in recs if "error" not in v["by_q"][q]]
        data = ("\n".join(v["line"] for v in good) + "\n").encode()
        last = {}
        for v in good:
            w = v["by_q"][q]
            last[(w["chrom"].encode(), w["pos"])] = w
        keys = sorted(last)
        _, res, ss = gpu_consensus(d, data, keys, [], p)
        for slot, key in enumerate(ss.key_tuples()):
            w = last[key]
            c = res.counts[slot]
            call = [x for x in w["calls"] if tuple(x["params"]) == params][0]
            assert c["status"] == L.ST_OK, w
            raw = int(c["raw_depth"])
            if not 0 <= w["raw"] < (1 << 32):                       # "-3", "5000000000": the value is in the position's spill record
                raw = int(res.spill[(int(c["n_symbols"]) >> 8) - 1]["depth64"])
                wide += 1
            assert (raw, c["good_depth"], c["fwd_good_depth"], c["rev_good_depth"]) == (w["raw"], w["good"], w["fwd"], w["rev"]), w
            assert chr(c["cons_base"]) == call["base"], (key, call)
            assert _names(int(c["filters"]), p) == call["failed"], (key, call)
            ranked = w["ranked"] or []
            assert c["n_symbols"] & 0xFF == len(ranked)
            tot, fw, rv = dict(map(tuple, w["total_hist"])), dict(map(tuple, w["fwd_hist"])), dict(map(tuple, w["rev_hist"]))
            for r, sym in enumerate(ranked[:L.MAX_SYMS]):
                assert chr(c["sym"][r]) == sym
                assert (c["total"][r], c["fwd"][r], c["rev"][r]) == (tot[sym], fw.get(sym, 0), rv.get(sym, 0)), (key, sym)
            if len(w["ref"]) > 1:                                   # the field itself is in the spill record, its first byte in the record
                more = res.spill[(int(c["n_symbols"]) >> 8) - 1]
                from snp_pipeline_amd.device import spill_reference_field
                assert spill_reference_field(res.spill, (int(c["n_symbols"]) >> 8) - 1).decode() == w["ref"] and chr(c["ref_base"]) == w["ref"][0]
                assert more["n"] == max(len(ranked) - L.MAX_SYMS, 0)
                long_refs += 1
            else:
                assert chr(c["ref_base"]) == w["ref"]
            if len(ranked) > L.MAX_SYMS:                            # the reference ranks any number of symbols: the rest is in the spill
                more = res.spill[(int(c["n_symbols"]) >> 8) - 1]
                assert more["n"] == len(ranked) - L.MAX_SYMS
                for r, sym in enumerate(ranked[L.MAX_SYMS:]):
                    assert (chr(more["sym"][r]), more["total"][r], more["fwd"][r], more["rev"][r]) == (sym, tot[sym], fw.get(sym, 0), rv.get(sym, 0)), (key, sym)
                spilled += 1
            checked += 1
        # the lane-per-site kernel on the same lines (no per-site counts requested)
        cons2, res2, _ = gpu_consensus(d, data, keys, [], p, want_counts=False)
        assert bytes(res2.bases) == bytes(res.bases) and bytes(res2.filters) == bytes(res.filters)
    _golden_records.wide_depths = wide
    return checked, spilled, long_refs


def test_golden_error_lines_raise(d, pileup_vectors):
    from snp_pipeline_amd.device import PileupFormatError
    from tests.gpu_util import gpu_consensus
    bad = [v for v in pileup_vectors["records"] if "error" in v["by_q"]["0"]]
    assert bad
    for v in bad[:40]:
        f = po.split_fields(v["line"].encode())
        with pytest.raises(PileupFormatError):
            gpu_consensus(d, (v["line"] + "\n").encode(), [(f[0], int(f[1]))], [], po.CallerParams())


def test_golden_whole_file_runs(d, pileup_vectors):
    """Whole files that went through the reference's own call_consensus driver: the FASTA string, from both device paths."""
    from tests.conftest import load_golden
    from tests.gpu_util import gpu_consensus
    # (pileup_runs3: the same through the reference's text-mode reader with CR LF, mixed and '\v' / '\f' line ends)
    for run in pileup_vectors["runs"] + load_golden("pileup_runs2.json.gz")["runs"] + load_golden("pileup_runs3.json.gz")["runs"]:
        kw = dict(run["kw"])
        if "contigs" in kw:
            kw["contigs"] = tuple(kw["contigs"])
        data, _, _ = fuzz.synth_pileup(run["seed"], **kw)
        if run.get("line_ends"):
            data = fuzz.with_line_ends(data, run["line_ends"], run["seed"])
        snps = [(c.encode(), p) for c, p in run["snplist"]]
        excl = [(c.encode(), p) for c, p in run["excluded"]]
        for want_counts in (True, False):
            cons, _, _ = gpu_consensus(d, data, snps, excl, po.CallerParams(*run["params"]), want_counts=want_counts)
            assert cons.decode() == run["consensus"], (run["seed"], want_counts)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_lines_vs_oracle(d, seed):
    """Fresh fuzz (not in the golden file), incl. long lines that span many 64-byte chunks."""
    from tests.gpu_util import check_against_oracle
    rng = random.Random(seed)
    lines, keys = [], []
    pos = 0
    while len(lines) < 1500:
        ln = fuzz.fuzz_line(rng)
        f = po.split_fields(ln.encode())
        try:
            po.parse_record(f, 0)
        except (IndexError, ValueError):
            continue
        pos += 1
        f[1] = str(pos).encode()                     # unique ascending positions
        lines.append(b"\t".join(f))
        keys.append((f[0], pos))
    data = b"\n".join(lines) + b"\n"
    for p in (po.CallerParams(), po.CallerParams(15, 0.9, 5, 2, 0.1), po.CallerParams(30, 0.75, 2, 1, 0.25)):
        check_against_oracle(d, data, keys, keys[::9], p)


def _long_line(rng, chrom, pos, depth):
    toks = "".join(fuzz._bases_token(rng) for _ in range(depth))
    approx = sum(1 for c in toks if c in ".,ACGTNacgtn*#<>")
    quals = "".join(chr(rng.randint(33, 74)) for _ in range(approx + rng.choice([0, 0, -3, 5])))
    return ("%s\t%d\t%s\t%d\t%s\t%s" % (chrom, pos, rng.choice("ACGTn"), depth, toks, quals)).encode()


def test_long_lines_both_paths(d):
    """Bases fields around and beyond the LDS limit (2048 B): wave-parallel and serial device paths."""
    from tests.gpu_util import check_against_oracle
    rng = random.Random(99)
    lines, keys = [], []
    for i, depth in enumerate([60, 63, 64, 65, 127, 128, 129, 500, 900, 1100, 1300, 1500, 2500, 6000, 20000]):
        lines.append(_long_line(rng, "deep", i + 1, depth))
        keys.append((b"deep", i + 1))
    data = b"\n".join(lines) + b"\n"
    check_against_oracle(d, data, keys, [], po.CallerParams(10, 0.6, 3, 0, 0.0))


def test_terminators_blank_and_missing_lines(d):
    from snp_pipeline_amd.device import PileupFormatError
    from tests.gpu_util import check_against_oracle, gpu_consensus
    body = [b"c1\t5\tA\t3\t..,\tIII", b"c1\t6\tC\t3\tTTt\tIII", b"c2\t5\tG\t2\t.$,\tII", b"c1\t9\tT\t0\t*\t*"]
    keys = [(b"c1", 5), (b"c1", 6), (b"c1", 7), (b"c2", 5), (b"c1", 9), (b"zz", 1)]
    for sep, tail in ((b"\n", b"\n"), (b"\r\n", b"\r\n"), (b"\r", b"\r"), (b"\n", b""), (b"\r\n", b"")):
        check_against_oracle(d, sep.join(body) + tail, keys, [(b"c1", 6)], po.CallerParams())
    # duplicate position: the last line wins (call_consensus.py:171-176)
    dup = b"c1\t5\tA\t3\tGGG\tIII\nc1\t5\tA\t3\tTTT\tIII\n"
    cons, _, _ = gpu_consensus(d, dup, [(b"c1", 5)], [], po.CallerParams())
    assert cons == b"T"
    # empty pileup, empty site list
    cons, _, _ = gpu_consensus(d, b"", keys, [], po.CallerParams())
    assert cons == b"-" * len(keys)
    cons, res, _ = gpu_consensus(d, body[0] + b"\n", [], [], po.CallerParams())
    assert cons == b"" and res.n_lines == 1
    # blank line / one-field line / non-numeric position make the reference raise ValueError (pileup.py:425-426)
    for bad in (b"c1\t5\tA\t1\t.\tI\n\nc1\t6\tA\t1\t.\tI\n", b"c1\n", b"c1\tx5\tA\t1\t.\tI\n", b"   \n"):
        with pytest.raises(PileupFormatError):
            gpu_consensus(d, bad, keys, [], po.CallerParams())


@pytest.mark.parametrize("seed,kw", [(21, dict(genome_len=6000, n_sites=150)),
                                      (22, dict(genome_len=3000, n_sites=90, contigs=("NODE_2", "NODE_10", "NODE_1", "N"))),
                                      (23, dict(genome_len=20000, n_sites=400, mean_depth=12)),
                                      # bases fields of 64..128 bytes (two mask words in the lane kernel) and lines past
                                      # its 256-byte window (wave-per-site kernel)
                                      (24, dict(genome_len=2500, n_sites=120, mean_depth=70)),
                                      (25, dict(genome_len=1500, n_sites=100, mean_depth=140))])
def test_synthetic_files_vs_oracle(d, seed, kw):
    from tests.gpu_util import check_against_oracle
    data, _, sites = fuzz.synth_pileup(seed, **kw)
    rng = random.Random(seed)
    # plus the positions around every power of ten (the scan's one-window parse recalibrates its digit count there)
    edges = [(sites[0][0], p10 + dlt) for p10 in (10, 100, 1000, 10000) for dlt in (-1, 0, 1)]
    snps = sorted(set(sites + edges + [(sites[0][0], 99_999_999), (b"absent_contig", 3)]))
    excl = rng.sample(sites, len(sites) // 5) + [(sites[0][0], 17)]
    res = check_against_oracle(d, data, snps, excl, po.CallerParams(0, 0.6, 3, 0, 0.0))
    assert res.n_lines == data.count(b"\n")
    check_against_oracle(d, data, snps, [], po.CallerParams(15, 0.9, 5, 2, 0.1))


def test_depth_sum_byproduct(d):
    """The sum of the depth column comes out of the same scan (fast path: 1..4 digit depths after a one-byte reference
    field; everything else through the exact parser), for the metrics step that re-reads the pileup in the reference."""
    from snp_pipeline_amd import device as dev
    data, _, sites = fuzz.synth_pileup(5, genome_len=5000, n_sites=20)
    ss = d.siteset(sites, [1] * len(sites))
    res = d.call_consensus(ss, data, dev.make_params(), want_depth_sum=True)
    assert res.depth_sum == po.depth_sum(data) > 0
    plain = d.call_consensus(ss, data, dev.make_params())
    assert bytes(plain.bases) == bytes(res.bases) and (plain.n_lines, plain.n_matched) == (res.n_lines, res.n_matched)
    # odd lines (fuzz.odd_depth_lines: depths of 5+ digits, two- and three-field lines, a multi-byte reference field, doubled
    # separators, a depth that is not a number, CR LF endings) — the very texts the reference's collect_metrics() summed for
    # tests/golden/metrics_vectors.json.gz, which pins the oracle
    n_odd = 4000
    for eol in (b"\n", b"\r\n"):
        odd = fuzz.odd_depth_lines(9, eol, n_odd)
        keys = [(b"c9", p_) for p_ in range(5, 4000, 37)]
        ss2 = d.siteset(keys, [1] * len(keys))
        got = d.call_consensus(ss2, odd, dev.make_params(), want_depth_sum=True, want_counts=False, check=False)
        assert got.depth_sum == po.depth_sum(odd), eol
        assert got.n_lines == n_odd


def test_device_pointer_api_unaligned_and_batch(d):
    """Device-resident pileups at odd offsets inside one buffer, batch entry point, torch stream."""
    import torch
    from snp_pipeline_amd import device as dev
    from snp_pipeline_amd import _lib as L
    d.use_torch_stream()
    samples = [fuzz.synth_pileup(30 + i, genome_len=4000, n_sites=70)[0] for i in range(3)]
    _, _, sites = fuzz.synth_pileup(30, genome_len=4000, n_sites=70)
    keys = sorted(set(sites) | {(b"synth_chr1", p) for p in range(1, 4000, 37)})
    ss = d.siteset(keys, [L.SITE_IN_SNPLIST] * len(keys))
    offs = [3]
    for s in samples:
        offs.append(offs[-1] + len(s) + 5)           # odd gaps => unaligned starts
    blob = np.full(offs[-1] + 64, ord("#"), dtype=np.uint8)
    starts = []
    for i, s in enumerate(samples):
        blob[offs[i]:offs[i] + len(s)] = np.frombuffer(s, dtype=np.uint8)
        starts.append(offs[i])
    t = torch.from_numpy(blob).cuda()
    n = len(ss)
    p = po.CallerParams(0, 0.6, 3, 0, 0.0)
    prm = dev.make_params(0, 0.6, 3, 0, 0.0)
    for i, s in enumerate(samples):
        bases = torch.empty(n, dtype=torch.uint8, device="cuda")
        filt = torch.empty(n, dtype=torch.uint8, device="cuda")
        status = torch.empty(4, dtype=torch.int64, device="cuda")
        d.call_consensus_dev(ss, t.data_ptr() + starts[i], len(s), prm, bases.data_ptr(), filt.data_ptr(), status.data_ptr())
        torch.cuda.synchronize()
        want, _ = po.call_consensus_sites(s, ss.key_tuples(), set(), p)
        assert bytes(bases.cpu().numpy()) == want
        assert int(status[0].item()) == -1
    # contiguous batch
    cat = np.concatenate([np.frombuffer(s, dtype=np.uint8) for s in samples])
    boffs = np.cumsum([0] + [len(s) for s in samples]).astype(np.uint64)
    tc = torch.from_numpy(cat).cuda()
    bases = torch.empty((3, n), dtype=torch.uint8, device="cuda")
    filt = torch.empty((3, n), dtype=torch.uint8, device="cuda")
    status = torch.empty((3, 4), dtype=torch.int64, device="cuda")
    d.call_consensus_batch_dev(ss, tc.data_ptr(), boffs, prm, bases.data_ptr(), filt.data_ptr(), status.data_ptr())
    torch.cuda.synchronize()
    for i, s in enumerate(samples):
        want, _ = po.call_consensus_sites(s, ss.key_tuples(), set(), p)
        assert bytes(bases[i].cpu().numpy()) == want
    d.sync()


def test_device_generated_pileup_vs_oracle(d):
    """The on-device generator feeds both the HIP path and the oracle (bytes copied back)."""
    import torch
    from snp_pipeline_amd import device as dev
    from snp_pipeline_amd import _lib as L
    G = 120_000
    ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda")
    d.use_torch_stream()
    d.synth_reference_dev(7, G, ref.data_ptr())
    torch.cuda.synchronize()
    refh = ref.cpu().numpy()
    rng = np.random.default_rng(3)
    pos = np.sort(rng.choice(np.arange(501, G - 500), size=1500, replace=False))
    alt = np.zeros(G + 1, dtype=np.uint8)
    for p_ in pos:
        alt[p_] = rng.choice([b for b in b"ACGT" if b != refh[p_]])
    alt_d = torch.from_numpy(alt).cuda()
    for sample in (0, 13):
        nbytes = d.synth_pileup_dev(11, sample, G, ref.data_ptr(), alt_d.data_ptr(), 0, 0)
        out = torch.empty(nbytes + 16, dtype=torch.uint8, device="cuda")
        n2 = d.synth_pileup_dev(11, sample, G, ref.data_ptr(), alt_d.data_ptr(), out.data_ptr(), nbytes + 16)
        assert n2 == nbytes
        data = bytes(out[:nbytes].cpu().numpy())
        assert data.count(b"\n") > 0.9 * G and 60 < nbytes / G < 120
        keys = [(b"synth_chr1", int(p_)) for p_ in pos]
        from tests.gpu_util import check_against_oracle
        check_against_oracle(d, data, keys, keys[::11], po.CallerParams(0, 0.6, 3, 0, 0.0))


def test_contig_changes_and_contigs_outside_the_site_set(d):
    """A genome-wide pileup walks through every contig; the scan kernel's per-wave contig hint must follow it —
    contigs with sites, contigs without any site, names longer than the 16-byte register compare and names longer
    than the hint itself (those lines take the exact parser)."""
    from snp_pipeline_amd.device import PileupFormatError
    from tests.gpu_util import check_against_oracle, gpu_consensus
    names = ("ctgA", "ctg_without_sites_1", "NODE_17_length_48211_cov_31.5", "b", "ctg_without_sites_2",
             "x" * 60, "ctgA2")
    data, _, sites = fuzz.synth_pileup(31, genome_len=2500, n_sites=40, contigs=names)
    snps = [k for k in sites if b"without" not in k[0]]
    res = check_against_oracle(d, data, snps, snps[::7], po.CallerParams(0, 0.6, 3, 0, 0.0))
    assert res.n_lines == data.count(b"\n")
    # no site at all on the contigs of the file / an empty site set
    check_against_oracle(d, data, [(b"elsewhere", 5)], [], po.CallerParams())
    cons, res, _ = gpu_consensus(d, data, [], [], po.CallerParams())
    assert cons == b"" and res.n_lines == data.count(b"\n")
    # a malformed position on a contig without sites still raises, as pileup.py:426 does for every line
    bad = data.replace(b"ctg_without_sites_2\t1200\t", b"ctg_without_sites_2\t12o0\t")
    assert bad != data
    with pytest.raises(PileupFormatError):
        gpu_consensus(d, bad, snps, [], po.CallerParams())


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_lane_kernel_alphabet_stress(d, seed):
    """Dense carets, indel markers, '$', digits and short / long quality strings over the alphabet the lane-per-site
    kernel keeps for itself (other symbols go to the wave-per-site kernel): both device paths against the oracle."""
    from tests.gpu_util import check_against_oracle
    rng = random.Random(seed)
    alphabet = ".,.,.,ACGTNacgtn*" + "^$+-0123456789"
    lines, keys = [], []
    for pos in range(1, 1501):
        # up to 128 bases: 256-byte window, two mask words; up to 255: 512-byte window, four mask words; more: wave kernel
        n = rng.choice([1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 120, 128, 129, 150, 192, 200, 240, 255, 256, 300])
        dense = rng.random() < 0.5
        toks = []
        while len(toks) < n:
            r = rng.random()
            if dense and r < 0.15:
                toks.append("^" * rng.randint(1, 4))
            elif dense and r < 0.30:
                k = rng.randint(0, 12)
                toks.append(rng.choice("+-") + (str(k) if rng.random() < 0.8 else "") + "".join(rng.choice("ACGTNacgtn") for _ in range(rng.randint(0, 4))))
            elif dense and r < 0.36:
                toks.append(str(rng.randint(0, 99)))
            elif r < 0.45 and dense:
                toks.append("$")
            else:
                toks.append(rng.choice(alphabet if dense else ".,.,.,.,ACGTacgt^$"))
        bases = "".join(toks)[:n]
        qn = max(0, len(bases) + rng.choice([0, 0, 0, -3, 2, -len(bases)]))
        quals = "".join(chr(33 + rng.choice([0, 5, 14, 15, 16, 30, 40, 41])) for _ in range(qn))
        ref = rng.choice("ACGTNacgtn")
        depth = rng.choice([len(bases), 1, 0]) if rng.random() < 0.2 else len(bases)
        fields = ["ctgL", str(pos), ref, str(depth), bases] + ([quals] if quals or rng.random() < 0.5 else [])
        lines.append(rng.choice(["\t", "\t", " "]).join(fields))
        keys.append((b"ctgL", pos))
    data = ("\n".join(lines) + "\n").encode()
    for p in (po.CallerParams(0, 0.6, 1, 0, 0.0), po.CallerParams(15, 0.75, 3, 1, 0.25), po.CallerParams(16, 0.5, 1, 0, 0.0)):
        ok_keys = []
        for (_, ln), k in zip(po.iter_lines(data), keys):       # keep the lines the reference itself can parse
            try:
                po.parse_record(po.split_fields(ln), p.min_base_quality)
                ok_keys.append(k)
            except Exception:
                pass
        assert len(ok_keys) > 1000
        check_against_oracle(d, data, ok_keys, ok_keys[::13], p)


def test_full_size_samples_properties_and_spot_check(d):
    """BASELINE configs[3] sample shape (5 Mbp x 30x, 50 k sites) — too big for the oracle, so: size-independent
    properties (line count = newline count, matched count = sites with a line, batch = per-sample = counts path,
    idempotence) plus the oracle on the very lines the device picked for 600 random sites."""
    import torch
    from snp_pipeline_amd import device as dev
    from snp_pipeline_amd import _lib as L
    G, S, B = 5_000_000, 50_000, 2
    d.use_torch_stream()
    ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda")
    d.synth_reference_dev(1, G, ref.data_ptr())
    refh = ref.cpu().numpy()
    rng = np.random.default_rng(2)
    pos = np.sort(rng.choice(np.arange(501, G - 499), size=S, replace=False))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    alt_h = np.zeros(G + 1, dtype=np.uint8)
    alt_h[pos] = acgt[(np.searchsorted(acgt, refh[pos]) + 1 + rng.integers(0, 3, size=S)) % 4]
    alt = torch.from_numpy(alt_h).cuda()
    sizes = [d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), 0, 0) for i in range(B)]
    offs = np.zeros(B, dtype=np.uint64)
    offs[1] = (sizes[0] + 255) // 256 * 256 + 7                       # an odd start address for the second sample
    pile = torch.full((int(offs[1]) + sizes[1] + 64,), 0x58, dtype=torch.uint8, device="cuda")
    for i in range(B):
        assert d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), pile.data_ptr() + int(offs[i]), sizes[i]) == sizes[i]
    keys = [(b"synth_chr1", int(p_)) for p_ in pos]
    ss = d.siteset(keys, [L.SITE_IN_SNPLIST] * S)
    p = po.CallerParams(0, 0.6, 3, 0, 0.0)
    prm = dev.make_params(p.min_base_quality, p.min_cons_freq, p.min_cons_depth, p.min_cons_strand_depth, p.min_cons_strand_bias)
    bases = torch.zeros((B, S), dtype=torch.uint8, device="cuda")
    filt = torch.zeros((B, S), dtype=torch.uint8, device="cuda")
    status = torch.zeros((B, 4), dtype=torch.int64, device="cuda")
    d.call_consensus_batch_dev(ss, pile.data_ptr(), offs, prm, bases.data_ptr(), filt.data_ptr(), status.data_ptr(),
                               sizes=np.asarray(sizes, dtype=np.uint64))
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    assert (st[:, 0] == -1).all()
    for i in range(B):
        view = pile[int(offs[i]):int(offs[i]) + sizes[i]]
        assert int(st[i, 1]) == int((view == 10).sum().item())          # lines seen = terminators in the file
    # the same samples one at a time: plain path and counts path; rerun of the batch (idempotence)
    for i in range(B):
        b1 = torch.zeros(S, dtype=torch.uint8, device="cuda")
        f1 = torch.zeros(S, dtype=torch.uint8, device="cuda")
        s1 = torch.zeros(4, dtype=torch.int64, device="cuda")
        d.call_consensus_dev(ss, pile.data_ptr() + int(offs[i]), sizes[i], prm, b1.data_ptr(), f1.data_ptr(), s1.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(b1, bases[i]) and torch.equal(f1, filt[i])
        assert s1.cpu().numpy().tolist() == st[i].tolist()
        line_off = d.line_offsets(ss)
        assert int(st[i, 2]) == int(np.count_nonzero(line_off))         # matched lines = sites that got a line
        assert ((bases[i].cpu().numpy() == 0x2D) | (line_off != 0)).all()
        cbuf = torch.zeros(S * 128, dtype=torch.uint8, device="cuda")
        b2 = torch.zeros(S, dtype=torch.uint8, device="cuda")
        f2 = torch.zeros(S, dtype=torch.uint8, device="cuda")
        d.call_consensus_dev(ss, pile.data_ptr() + int(offs[i]), sizes[i], prm, b2.data_ptr(), f2.data_ptr(), s1.data_ptr(),
                             d_counts=cbuf.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(b2, bases[i]) and torch.equal(f2, filt[i])   # wave-per-site kernel = lane-per-site kernel
        # the oracle on the lines the device picked
        bh, fh = bases[i].cpu().numpy(), filt[i].cpu().numpy()
        for slot in rng.choice(S, size=300, replace=False):
            if line_off[slot] == 0:
                assert bh[slot] == 0x2D and fh[slot] == 0
                continue
            a0 = int(offs[i]) + int(line_off[slot]) - 1
            raw = bytes(pile[a0:a0 + 700].cpu().numpy())
            line = raw[:raw.index(b"\n")]
            if a0 > int(offs[i]):
                assert int(pile[a0 - 1].item()) == 10                    # it is the start of a line
            fields = po.split_fields(line)
            assert (fields[0], int(fields[1])) == keys[slot]
            base, mask = po.call_record(po.parse_record(fields, p.min_base_quality), p)
            want = 0x2D if (mask or base == 0x2A) else base
            assert (int(bh[slot]), int(fh[slot])) == (want, mask), line
    b3 = torch.zeros((B, S), dtype=torch.uint8, device="cuda")
    f3 = torch.zeros((B, S), dtype=torch.uint8, device="cuda")
    d.call_consensus_batch_dev(ss, pile.data_ptr(), offs, prm, b3.data_ptr(), f3.data_ptr(), status.data_ptr(),
                               sizes=np.asarray(sizes, dtype=np.uint64))
    torch.cuda.synchronize()
    assert torch.equal(b3, bases) and torch.equal(f3, filt)


def test_batch_groups_empty_and_tiny_samples(d):
    """More samples than one scan launch takes (groups of 256), with empty, one-line and unterminated samples mixed in,
    at odd offsets inside one buffer: every row equals the per-sample host call and the oracle."""
    import torch
    from snp_pipeline_amd import device as dev
    rng = random.Random(77)
    base, _, sites = fuzz.synth_pileup(41, genome_len=400, n_sites=25)
    keys = sorted(sites)
    lines = base.split(b"\n")[:-1]
    samples = []
    for i in range(300):
        kind = i % 6
        if kind == 0:
            samples.append(b"")
        elif kind == 1:
            samples.append(rng.choice(lines) + b"\n")
        elif kind == 2:
            samples.append(b"\n".join(rng.sample(lines, 40)))               # no terminator at the end of the file
        else:
            pick = sorted(rng.sample(range(len(lines)), rng.randint(50, len(lines))))
            samples.append(b"\n".join(lines[j] for j in pick) + b"\n")
    p = po.CallerParams(0, 0.6, 3, 0, 0.0)
    prm = dev.make_params(p.min_base_quality, p.min_cons_freq, p.min_cons_depth, p.min_cons_strand_depth, p.min_cons_strand_bias)
    ss = d.siteset(keys, [1] * len(keys))
    offs, blob = [], bytearray()
    for sm in samples:
        blob += b"Z" * rng.randint(0, 5)                                     # junk between samples, odd alignment
        offs.append(len(blob))
        blob += sm
    blob += b"ZZZZ"
    t = torch.from_numpy(np.frombuffer(bytes(blob), dtype=np.uint8).copy()).cuda()
    n, S = len(samples), len(keys)
    bases = torch.zeros((n, S), dtype=torch.uint8, device="cuda")
    filt = torch.zeros((n, S), dtype=torch.uint8, device="cuda")
    status = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    d.use_torch_stream()
    d.call_consensus_batch_dev(ss, t.data_ptr(), np.asarray(offs, dtype=np.uint64), prm, bases.data_ptr(), filt.data_ptr(),
                               status.data_ptr(), sizes=np.asarray([len(sm) for sm in samples], dtype=np.uint64))
    torch.cuda.synchronize()
    bh, st = bases.cpu().numpy(), status.cpu().numpy()
    order = [ss.index_of[i] for i in range(S)]
    for i, sm in enumerate(samples):
        want, _ = po.call_consensus_sites(sm, keys, set(), p)
        got = bytes(int(bh[i, j]) for j in order)
        assert got == want, i
        assert st[i, 0] == -1 and st[i, 1] == len(list(po.iter_lines(sm))), i


def test_queue_overflow_takes_the_exact_pass(d):
    """More lines than the slow-line queue holds (1 M) are separated by two TABs, which only the exact parser accepts: the
    queue overflows and the exact instantiation of the scan redoes the file.  Counts, matches and calls must not change."""
    from tests.gpu_util import gpu_consensus
    n = 1_100_000
    rng = random.Random(5)
    keys = sorted({(b"ovf", rng.randrange(1, n + 1)) for _ in range(300)})
    pos = np.arange(1, n + 1)
    body = b"".join(b"ovf\t\t%d\tA\t2\t.,\tII\n" % p for p in range(1, n + 1))
    cons, res, ss = gpu_consensus(d, body, keys, [], po.CallerParams(0, 0.6, 1, 0, 0.0), want_counts=False)
    assert res.n_lines == n and res.n_matched == len(keys)
    assert cons == b"A" * len(keys)
    # the same file with ordinary separators (fast path) gives the same answer
    cons2, res2, _ = gpu_consensus(d, body.replace(b"\t\t", b"\t"), keys, [], po.CallerParams(0, 0.6, 1, 0, 0.0), want_counts=False)
    assert cons2 == cons and res2.n_lines == n and res2.n_matched == len(keys)
    # and a malformed position far into the overflowing file is still reported
    from snp_pipeline_amd.device import PileupFormatError
    bad = body[:-20] + body[-20:].replace(b"\t\t11", b"\t\t1x")
    assert bad != body
    with pytest.raises(PileupFormatError):
        gpu_consensus(d, bad, keys, [], po.CallerParams(0, 0.6, 1, 0, 0.0), want_counts=False)


def test_python_int_fields(d):
    """int(pos) / int(depth) as CPython parses them (pileup.py:426, :223-225): an optional sign and single underscores between
    digits are integers too; the one-window parse leaves such lines to the exact parser, which must agree with the oracle."""
    from snp_pipeline_amd.device import PileupFormatError
    from tests.gpu_util import check_against_oracle, gpu_consensus
    body = [b"c1\t+5\tA\t3\t..,\tIII", b"c1\t1_0\tC\t+3\tTTt\tIII", b"c1\t0_1_2\tG\t1_0\t.$,.,.,.,.,\tIIIIIIIIII", b"c1\t-0\tT\t2\tgg\tII",
             b"c1\t-7\tT\t2\tgg\tII", b"c1\t007\tT\t02\tcc\tII", b"c1\t99999999999\tA\t1\t.\tI", b"c1\t20\tA\t0_0\t*\t*"]
    keys = [(b"c1", 5), (b"c1", 10), (b"c1", 12), (b"c1", 0), (b"c1", 7), (b"c1", 20), (b"c1", 99)]
    data = b"\n".join(body) + b"\n"
    res = check_against_oracle(d, data, keys, [(b"c1", 10)], po.CallerParams(0, 0.6, 1, 0, 0.0))
    assert res.n_lines == len(body) and res.n_matched == 6
    got = d.call_consensus(d.siteset(keys, [1] * len(keys)), data, __import__("snp_pipeline_amd.device", fromlist=["x"]).make_params(),
                           want_depth_sum=True)
    assert got.depth_sum == po.depth_sum(data) == 3 + 3 + 10 + 2 + 2 + 2 + 1 + 0
    for bad in (b"c1\t5_\tA\t1\t.\tI\n", b"c1\t_5\tA\t1\t.\tI\n", b"c1\t+\tA\t1\t.\tI\n", b"c1\t5__0\tA\t1\t.\tI\n", b"c1\t++5\tA\t1\t.\tI\n",
                b"c1\t5\tA\t1_\t.\tI\n", b"c1\t5\tA\t+\t.\tI\n"):
        with pytest.raises(ValueError):
            po.call_consensus_sites(bad, keys, set(), po.CallerParams())
        with pytest.raises(PileupFormatError) as ei:
            gpu_consensus(d, bad, keys, [], po.CallerParams())
        assert ei.value.reference_exception is ValueError
    # a negative depth is an integer for the reference, which compares it with 0 and prints it: since round 4 the device does
    # the same (the value rides in the position's spill record); only a depth of 2^62 and beyond is still refused
    res = check_against_oracle(d, b"c1\t5\tA\t-3\t...\tIII\nc1\t7\tC\t5000000000\tGGg\tIII\n", keys, [], po.CallerParams(0, 0.6, 1, 0, 0.0))
    slots = {k: i for i, k in enumerate(sorted(keys))}
    for key, want in (((b"c1", 5), -3), ((b"c1", 7), 5000000000)):
        c = res.counts[slots[key]]
        assert int(res.spill[(int(c["n_symbols"]) >> 8) - 1]["depth64"]) == want and int(c["good_depth"]) == 3
    with pytest.raises(PileupFormatError):
        gpu_consensus(d, b"c1\t5\tA\t4611686018427387904\t...\tIII\n", keys, [], po.CallerParams())
    # fields of 20 to 25 digits: int() takes them whole, so such a position is in no site set — 2^64 + 4 must not come out as 4
    # (the accumulator saturates before its multiply) — and such a depth is refused like any other from 2^62 on
    for pos in (b"18446744073709551620", b"18446744073709551621", b"184467440737095516165", b"9999999999999999999999999", b"46116860184273879045",
                b"+18446744073709551620", b"18_446_744_073_709_551_620"):
        data = b"c1\t7\tC\t2\tGG\tII\nc1\t" + pos + b"\tA\t3\t...\tIII\nc1\t10\tC\t2\tTT\tII\n"
        res = check_against_oracle(d, data, [(b"c1", 4), (b"c1", 5), (b"c1", 7), (b"c1", 10)], [], po.CallerParams(0, 0.6, 1, 0, 0.0))
        assert res.n_lines == 3 and res.n_matched == 2
    for depth in (b"20000000000000000000", b"18446744073709551619", b"9999999999999999999999999"):
        with pytest.raises(PileupFormatError):
            gpu_consensus(d, b"c1\t5\tA\t" + depth + b"\t...\tIII\n", keys, [], po.CallerParams())
        got = d.call_consensus(d.siteset(keys, [1] * len(keys)), b"c1\t9\tA\t" + depth + b"\t...\tIII\n", __import__("snp_pipeline_amd.device", fromlist=["x"]).make_params())
        assert got.n_lines == 1 and got.n_matched == 0                  # (a line that is not listed is never converted: pileup.py:426-429)


def test_the_first_malformed_line_in_file_order_decides_the_exception(d, tmp_path):
    """The reference stops at the first line it cannot take, whichever kind: one whose position the reader cannot convert (any
    line, ValueError) or one at a listed position that Record cannot be built from (IndexError / ValueError).  Found by
    tools/fuzz_campaign.py: a lone CR inside the read bases of a listed line makes two lines, "c1 145 A 3 c" (IndexError if listed)
    and "cc JEJ" (ValueError) — the device used to report the reader-level one whatever its place."""
    from snp_pipeline_amd.device import PileupFormatError
    from tests.gpu_util import gpu_consensus
    good = b"".join(b"c1\t%d\tA\t3\t...\tIII\n" % k for k in range(1, 60))
    split = b"c1\t60\tA\t3\tc\rcc\tJEJ\n"
    tail = [b"c1\t%d\tA\t3\t...\tIII\n" % k for k in range(61, 90)]
    tail = [b"".join(tail[k:]) for k in range(len(tail))]           # tail[10:] starts at position 71
    bad_pos = b"c1\tx7\tA\t3\t...\tIII\n"
    five = b"c1\t70\tA\t3\t...\n"
    p = po.CallerParams()
    cases = [(good + split + tail[0], [(b"c1", 60)], IndexError),         # listed: Record fails first
             (good + split + tail[0], [(b"c1", 59)], ValueError),         # not listed: the reader fails on "cc JEJ"
             (good + bad_pos + five + tail[10], [(b"c1", 70)], ValueError),   # reader-level line first
             (good + five + bad_pos + tail[10], [(b"c1", 70)], IndexError),   # record-level line first
             (good + five + bad_pos + tail[10], [(b"c1", 5)], ValueError)]
    for data, keys, exc in cases:
        with pytest.raises(exc):
            po.call_consensus_sites(data, keys, set(), p)
        with pytest.raises(PileupFormatError) as ei:
            gpu_consensus(d, data, keys, [], p)
        assert ei.value.reference_exception is exc, (keys, exc)
        path = str(tmp_path / "f.pileup")
        with open(path, "wb") as f:
            f.write(data)
        ss = d.siteset(keys, [1] * len(keys))
        results, rcs, _ = d.call_consensus_files(ss, [path], devmod_params(p), want_counts=True, want_line_offsets=True)
        with pytest.raises(PileupFormatError) as ei:
            d.raise_file_status(path, int(rcs[0]), results[0])
        assert ei.value.reference_exception is exc, ("files", keys, exc)


def devmod_params(p):
    from snp_pipeline_amd import device as devmod
    return devmod.make_params(p.min_base_quality, p.min_cons_freq, p.min_cons_depth, p.min_cons_strand_depth, p.min_cons_strand_bias)


def test_site_sets_and_stores_outliving_their_device_are_closed_with_it():
    """A site set or a pileup store points into its context; closing the device first must not leave either to be
    destroyed against freed memory later (the garbage collector runs their __del__ whenever it likes)."""
    from snp_pipeline_amd import device as devmod
    dev = devmod.Device(0)
    ss = devmod.SiteSet.from_arrays(dev, [b"c1"], np.array([5, 9], dtype=np.uint64), np.array([1, 1], dtype=np.uint8))
    store = dev.pileups(1 << 20)
    dev.close()
    assert ss.handle is None and store.handle is None
    ss.close(), store.close()                               # and again: nothing left to do
    del ss, store
    again = devmod.Device(0)                                # the process' HIP state is intact
    try:
        ss2 = devmod.SiteSet.from_arrays(again, [b"c1"], np.array([5], dtype=np.uint64), np.array([1], dtype=np.uint8))
        assert len(ss2) == 1
    finally:
        again.close()


@pytest.mark.parametrize("names", [None, ("NODE_1_length_419034_cov_23.1", "gi|9626243|ref|NC_001416.1|_and_some_more_text", "c")])
@pytest.mark.parametrize("variant", ["crlf", "mixed", "vt_ff"])
def test_terminators_across_many_tiles(d, variant, names):
    """Files of many scan tiles whose lines end in "\\r\\n" (pairs inside and across the 16-byte chunks of the index), in a mix
    of "\\n", "\\r\\n" and lone "\\r", or carry '\\v' / '\\f' (flagged by the index, no line terminators): the fast parse sorts the
    flagged starts out itself, line counts and calls as the oracle's (Python's universal newlines)."""
    import io
    from tests.gpu_util import check_against_oracle
    # (names: contig names too long for the byte before the line to sit in the parse's 24-byte window — it is read separately —
    # and, beyond 22 - digits bytes, checked in two pieces)
    kw = dict(contigs=names) if names else {}
    data, _, sites = fuzz.synth_pileup(31, genome_len=30000 if not names else 12000, n_sites=500, mean_depth=20, **kw)
    rng = random.Random(7)
    lines = data.split(b"\n")[:-1]
    if variant == "crlf":
        data2 = b"\r\n".join(lines) + b"\r\n"
    elif variant == "mixed":
        data2 = b"".join(ln + rng.choice((b"\n", b"\n", b"\r\n", b"\r")) for ln in lines)
    else:
        data2 = b"".join(ln + rng.choice((b"", b"", b"\x0b", b"\x0c", b" \x0b")) + b"\n" for ln in lines)
    assert len(data2) > 40 * 4096
    snps = sorted(set(sites + [(sites[0][0], 99_999_999)]))
    res = check_against_oracle(d, data2, snps, rng.sample(sites, 50), po.CallerParams(0, 0.6, 3, 0, 0.0))
    n_py = sum(1 for _ in io.TextIOWrapper(io.BytesIO(data2), encoding="latin-1", newline=None))
    assert res.n_lines == n_py == len(lines)


def test_more_spilled_positions_than_the_first_arena_holds_grow_the_arena(d):
    """1 100 positions with nine symbols in one call: a fresh context keeps 1 024 spill records.  Round 3 refused the positions past
    that (they carry the "no room" mark in their own record); now the read of the spill says how many were asked for, the context
    allocates that many for its next call, and the call is repeated (device.SpillOverflow inside, invisible outside): every
    position has its record and its full ALT list."""
    from snp_pipeline_amd import _lib as L
    from snp_pipeline_amd import device as dev
    from snp_pipeline_amd import vcf_writer
    from tests.gpu_util import gpu_consensus
    bases = "ACGTN*RYK" * 2
    lines = ["c\t%d\tA\t18\t%s\t%s" % (p, bases, "I" * 18) for p in range(1, 1101)]
    keys = [(b"c", p) for p in range(1, 1101)]
    data = ("\n".join(lines) + "\n").encode()
    # the bare call on a context of its own: the first attempt runs out and says so, the second finds room
    d2 = dev.Device(0)
    try:
        ss2 = d2.siteset(keys, [L.SITE_IN_SNPLIST] * len(keys))
        assert d2.lib.snpgpu_symbol_spill_capacity(d2.ctx) == L.SPILL_CAP
        with pytest.raises(dev.SpillOverflow):
            dev.Device.call_consensus.__wrapped__(d2, ss2, data, dev.make_params(), want_counts=True)
        again = dev.Device.call_consensus.__wrapped__(d2, ss2, data, dev.make_params(), want_counts=True)
        assert len(again.spill) == 1100 and d2.lib.snpgpu_symbol_spill_capacity(d2.ctx) >= 1100
    finally:
        d2.close()
    _, res, ss = gpu_consensus(d, data, keys, [], po.CallerParams())
    codes = res.counts["n_symbols"] >> 8
    assert (res.counts["n_symbols"] & 0xFF == 9).all() and len(res.spill) == 1100
    assert sorted(codes) == list(range(1, 1101))
    names = po.filter_names(po.CallerParams())
    order = np.arange(1100, dtype=np.uint32)
    text = vcf_writer.format_rows(res.counts, order, ss._names, ss._offs, ss.keys, names, False, ".", spill=res.spill)
    assert text.count(b"\n") == 1100 and all(ln.split(b"\t")[4].count(b",") == 7 for ln in text.split(b"\n") if ln)   # A is REF: 8 ALTs
    with pytest.raises(ValueError):                              # (a record that points at a spill record the writer was not given)
        vcf_writer.format_rows(res.counts, order, ss._names, ss._offs, ss.keys, names, False, ".", spill=res.spill[:500])


def test_files_with_one_odd_line_end_as_the_reference_does(d):
    """The reference's driver on files with one odd line (badline_runs.json.gz): the same exception class — carried by
    PileupFormatError.reference_exception, which the CLI re-raises as — or the same consensus, with and without per-site records."""
    from snp_pipeline_amd.device import PileupFormatError
    from tests.conftest import load_golden
    from tests.gpu_util import gpu_consensus
    for run in load_golden("badline_runs.json.gz")["runs"]:
        kw = dict(run["kw"])
        if "contigs" in kw:
            kw["contigs"] = tuple(kw["contigs"])
        base, _, _ = fuzz.synth_pileup(run["seed"], **kw)
        snps = [(c.encode(), p) for c, p in run["snplist"]]
        data = fuzz.with_bad_line(base, run["scenario"], set(snps))
        for want_counts in (True, False):
            if "exception" in run:
                with pytest.raises(PileupFormatError) as ei:
                    gpu_consensus(d, data, snps, [], po.CallerParams(*run["params"]), want_counts=want_counts)
                assert ei.value.reference_exception.__name__ == run["exception"], (run["scenario"], want_counts)
            else:
                cons, _, _ = gpu_consensus(d, data, snps, [], po.CallerParams(*run["params"]), want_counts=want_counts)
                assert cons.decode() == run["consensus"], (run["scenario"], want_counts)


@pytest.mark.parametrize("seed", [3, 4])
def test_many_short_lines_per_lane_of_the_scan(d, seed):
    """The scan keeps no list of line starts: a lane parses the lines that start behind the terminators of its own 64 bytes, one per
    round.  Lines of 11 to 20 bytes put three to six of them into a lane's bytes (as many rounds per tile), stretches of deep lines
    in between change the sample's line density from tile to tile, and the first / last tiles are short ones; counts and calls as
    the oracle's (pileup.py:422-429)."""
    from tests.gpu_util import check_against_oracle
    rng = random.Random(seed)
    out, sites, pos = [], [], 0
    while sum(len(x) for x in out) < 30 * 4096:
        if rng.random() < 0.15:                                  # a stretch of deep lines
            for _ in range(rng.randrange(1, 40)):
                pos += rng.randrange(1, 3)
                dp = rng.randrange(20, 120)
                out.append(b"c\t%d\tA\t%d\t%s\t%s\n" % (pos, dp, bytes(rng.choice(b".,ACgt") for _ in range(dp)), b"I" * dp))
                if rng.random() < 0.3:
                    sites.append((b"c", pos))
        else:                                                    # a stretch of very short ones
            for _ in range(rng.randrange(5, 400)):
                pos += rng.randrange(1, 3)
                dp = rng.randrange(0, 3)
                out.append(b"c\t%d\tG\t%d\t%s\t%s\n" % (pos, dp, (b"." * dp) or b"*", (b"I" * dp) or b"*"))
                if rng.random() < 0.1:
                    sites.append((b"c", pos))
    data = b"".join(out)
    res = check_against_oracle(d, data, sorted(set(sites)), rng.sample(sites, 20), po.CallerParams(0, 0.6, 1, 0, 0.0))
    assert res.n_lines == len(out) and res.n_matched == len(set(sites))


@pytest.mark.parametrize("seed", [5, 6])
def test_every_line_a_site_and_positions_repeated_far_apart(d, seed):
    """The scan collects its matched lines in LDS (128 per wave) and publishes them with atomicMax a tile later.  A file whose
    every line is a site fills that buffer inside one tile (it is emptied in the middle of a round then), and a position that
    comes back tens of tiles later — in another wave's run — must still end with its LAST line (pileup.py:422-429 keeps the last
    record of a position): whatever the order in which the waves publish."""
    from tests.gpu_util import check_against_oracle
    rng = random.Random(seed)
    out, pos = [], 0
    while sum(len(x) for x in out) < 40 * 4096:
        pos += 1
        dp = rng.randrange(1, 4) if rng.random() < 0.8 else rng.randrange(10, 60)
        out.append(b"c\t%d\t%s\t%d\t%s\t%s\n" % (pos, bytes([rng.choice(b"ACGT")]), dp, bytes(rng.choice(b".,ACgt") for _ in range(dp)), b"I" * dp))
    n_first = len(out)
    for _ in range(300):                                         # old positions again, with other bases, far behind their first lines
        q = rng.randrange(1, pos + 1)
        dp = rng.randrange(3, 9)
        out.append(b"c\t%d\tA\t%d\t%s\t%s\n" % (q, dp, bytes(rng.choice(b"CGT") for _ in range(dp)), b"I" * dp))
    data = b"".join(out)
    sites = [(b"c", q) for q in range(1, pos + 1)]
    res = check_against_oracle(d, data, sites, rng.sample(sites, 50), po.CallerParams(0, 0.6, 1, 0, 0.0))
    assert res.n_lines == len(out) and res.n_matched == len(out) and n_first == pos


@pytest.mark.parametrize("order", ["shuffled", "blocks", "zero_padded"])
def test_positions_whose_top_digits_keep_changing(d, order):
    """The one-window parse of the scan knows all but the last four digits of a position in advance; a round with a line that differs
    is done again in the general form, and after four such rounds in a row the wave stays with the general form for a while.  Files
    that are not sorted (every line another range), sorted in short blocks that jump back and forth across 10^4 and digit-count
    boundaries, and positions written with leading zeros: counts and calls as the oracle's (pileup.py:422-429 knows no order)."""
    from tests.gpu_util import check_against_oracle
    rng = random.Random({"shuffled": 11, "blocks": 12, "zero_padded": 13}[order])
    n = 9000
    if order == "shuffled":
        positions = rng.sample(range(1, 3_000_000), n)
    elif order == "blocks":
        positions = []
        while len(positions) < n:
            start = rng.choice((1, 95, 990, 9_980, 19_990, 99_985, 123_456, 999_990, 1_239_990, 42_949_600, 4_294_967_270))
            positions += list(range(start, start + rng.randrange(5, 60)))
    else:
        positions = list(range(9_900, 9_900 + n))
    lines, sites = [], []
    for p in positions:
        dp = rng.randrange(1, 40)
        ptxt = (b"%09d" % p) if order == "zero_padded" and rng.random() < 0.7 else b"%d" % p
        lines.append(b"chrT\t%s\tC\t%d\t%s\t%s\n" % (ptxt, dp, bytes(rng.choice(b".,Aa") for _ in range(dp)), b"F" * dp))
        if rng.random() < 0.05 and p < 100_000_000:               # (positions beyond 2^32 - 1 are lines like any other, in no site set)
            sites.append((b"chrT", p))
    data = b"".join(lines)
    assert len(data) > 40 * 4096
    snps = sorted(set(sites))
    res = check_against_oracle(d, data, snps, rng.sample(snps, 10), po.CallerParams(0, 0.6, 1, 0, 0.0))
    assert res.n_lines == len(lines)


@pytest.mark.parametrize("depth_sum", [False, True])
def test_many_contig_changes_inside_one_scan_tile(d, depth_sum):
    """Round 6: the scan puts the lines of another contig than its hint aside and takes them in another trip through the tile after
    the contig change, as often as the tile changes contig (up to round 5 they went to the exact parser).  Contigs of five to forty
    lines — a dozen changes per 4 KiB tile — with names the hint cannot hold (50 bytes: those lines do go to the exact parser), names
    that are not in the site set, the same name coming back later (an unsorted file), a name followed by two blanks, and a contig whose
    lines fill several tiles in between; consensus, line counts and the depth-column sum as the oracle's."""
    from snp_pipeline_amd import _lib as L
    from tests.gpu_util import check_against_oracle
    rng = random.Random(77)
    names = ["c%02d" % i for i in range(30)] + ["NODE_%d_length_%d_cov_%.1f" % (i, 1000 + i, 3.5 + i) for i in range(12)] + ["L" * 50, "absent_1", "absent_2"]
    lines, keys = [], []
    order = names + ["c03", "bulk", "c07", "L" * 50, "c01"]                     # (names that come a second time)
    for k, name in enumerate(order):
        n_lines = 400 if name == "bulk" else rng.choice((3, 5, 9, 17, 40))
        first = 100 * k + 1 if name != "bulk" else 1
        for j in range(n_lines):
            pos = first + j
            depth = rng.randint(3, 25)
            reads = "".join(rng.choice("..,,..,,AaCcGgTt") for _ in range(depth))
            sep = "  " if (name == "c05" and j == 2) else "\t"               # the hint's own name with something odd behind it
            lines.append("%s%s%d\t%s\t%d\t%s\t%s" % (name, sep, pos, rng.choice("ACGT"), depth, reads, "I" * depth))
            if not name.startswith("absent") and j % 3 == 0:
                keys.append((name.encode(), pos))
    data = ("\n".join(lines) + "\n").encode()
    assert len(order) > 3 * (len(data) // 4096)                               # several contig changes per 4 KiB tile
    snps = sorted(set(keys))
    p = po.CallerParams(0, 0.6, 3, 0, 0.0)
    res = check_against_oracle(d, data, snps, snps[::11], p)
    assert res.n_lines == len(lines)
    if depth_sum:
        ss = d.siteset(snps, [L.SITE_IN_SNPLIST] * len(snps))
        r2 = d.call_consensus(ss, data, devmod_params(p), want_counts=False, want_depth_sum=True)
        assert r2.depth_sum == po.depth_sum(data)


def test_an_earlier_malformed_line_hidden_behind_a_later_line_of_its_position(d, tmp_path):
    """Found by tools/fuzz_campaign.py in round 6 (seed 867783): a listed position that comes three times — well-formed, then with
    three fields (IndexError), then with a depth of 'C' (ValueError) — and a line with a bad position column between the last two.
    The per-site result knows only the LAST line of the position (ValueError, behind the reader-level ValueError), the reference ends
    at the first of them all (IndexError): Device.raise_file_errors lets the all-lines pass decide whenever a listed position repeats."""
    from snp_pipeline_amd.device import PileupFormatError
    good = b"".join(b"c1\t%d\tA\t3\t...\tIII\n" % k for k in range(1, 60))
    tail = b"".join(b"c1\t%d\tA\t3\t...\tIII\n" % k for k in range(61, 90))
    three = b"c1\t5\tA\n"                      # position 5 again: IndexError when listed
    bad_depth = b"c1\t5\t8\tC\t1\tg\tJ\n"      # ... and again: ValueError when listed
    bad_pos = b"12\tT\t9\t,,,,....,\tDJH@ED?BC\n"
    again = b"c1\t5\tG\t2\t..\tII\n"           # ... and a well-formed last one
    p = po.CallerParams()
    cases = [(good + three + bad_pos + bad_depth + tail, [(b"c1", 5)], IndexError),
             (good + bad_depth + bad_pos + three + tail, [(b"c1", 5)], ValueError),
             (good + bad_pos + three + bad_depth + tail, [(b"c1", 5)], ValueError),       # the reader-level line first
             (good + three + bad_depth + again + tail, [(b"c1", 5)], IndexError),         # no reader-level error, the last line of the position is fine
             (good + three + bad_pos + bad_depth + tail, [(b"c1", 7)], ValueError),       # position 5 not listed: only the reader-level line counts
             (good + three + again + tail, [(b"c1", 5), (b"c1", 9)], IndexError)]
    for k, (data, keys, exc) in enumerate(cases):
        with pytest.raises(exc):
            po.call_consensus_sites(data, keys, set(), p)
        path = str(tmp_path / ("f%d.pileup" % k))
        with open(path, "wb") as f:
            f.write(data)
        ss = d.siteset(keys, [1] * len(keys))
        results, rcs, _ = d.call_consensus_files(ss, [path], devmod_params(p), want_counts=True, want_line_offsets=True)
        with pytest.raises(PileupFormatError) as ei:
            d.raise_file_errors(ss, path, devmod_params(p), int(rcs[0]), results[0])
        assert ei.value.reference_exception is exc, (k, exc, ei.value)
    # ... and the same through the console script's function: the exception class the reference ends with
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    data, keys, exc = cases[0]
    sdir = tmp_path / "s"
    sdir.mkdir()
    (sdir / "reads.all.pileup").write_bytes(data)
    (tmp_path / "snplist.txt").write_text("c1\t5\t1\ts\n")
    a = cli.parse_argument_list(("call_consensus -v 0 -f -l %s/snplist.txt -o %s/consensus.fasta %s/reads.all.pileup" % (tmp_path, sdir, sdir)).split())
    with pytest.raises(exc):
        a.func(a)

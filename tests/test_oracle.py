"""The CPU oracle against vectors produced by the real reference
(oracle/gen_golden.py) and against the reference's bundled ExpectedResults."""
import os

import pytest

from oracle import pileup_oracle as po
from oracle import steps_oracle as so


def test_reference_doctests_were_green(pileup_vectors):
    d = pileup_vectors["reference_doctests"]
    assert d["failed"] == 0 and d["attempted"] >= 79


def test_strip_vectors(pileup_vectors):
    for raw, want in pileup_vectors["strip"]:
        assert po.strip_bases(raw.encode()).decode() == want, raw


def _hist(d):
    return sorted([chr(k), v] for k, v in d.items())


def test_record_and_caller_vectors(pileup_vectors):
    assert _check_records(pileup_vectors["records"]) > 15000


def test_reference_fields_of_several_bytes(longref_vectors):
    """pileup.py:223 takes any string as the reference base; '.' / ',' then stand for the whole upper- / lower-cased field
    (pileup.py:255-258): counts, ranking and calls of the real reference on ~600 such lines."""
    recs = longref_vectors["records"]
    assert all(len(v["line"].split("\t")[2]) > 1 for v in recs)
    assert _check_records(recs) > 1500
    # ... and on lines whose depth column is an integer outside 0 .. 2^32 - 1 ("-3", "5000000000", "1_000": pileup.py:225)
    wide = longref_vectors["wide_depth_records"]
    assert _check_records(wide) > 60 and sum(1 for v in wide if not 0 <= v["by_q"]["0"]["raw"] < (1 << 32)) >= 15
    # ... and whose reference field is longer than one spill record of the device holds (65 bytes to 20 000)
    very = longref_vectors["very_long_ref_records"]
    assert _check_records(very) > 100 and max(len(v["line"].split("\t")[2]) for v in very) == 20000


def _check_records(records):
    n_calls = 0
    for v in records:
        line = v["line"].encode()
        for q, want in v["by_q"].items():
            q = int(q)
            if "error" in want:
                with pytest.raises((IndexError, ValueError)) as ei:
                    po.parse_record(po.split_fields(line), q)
                assert type(ei.value).__name__ == want["error"], v["line"]
                continue
            r = po.parse_record(po.split_fields(line), q)
            assert (r.chrom.decode(), r.position, r.reference_base.decode(), r.raw_depth) == \
                (want["chrom"], want["pos"], want["ref"], want["raw"]), v["line"]
            assert (r.good_depth, r.forward_good_depth, r.reverse_good_depth) == (want["good"], want["fwd"], want["rev"]), v["line"]
            assert _hist(r.base_good_depth) == want["total_hist"], v["line"]
            assert _hist(r.forward_base_good_depth) == want["fwd_hist"], v["line"]
            assert _hist(r.reverse_base_good_depth) == want["rev_hist"], v["line"]
            ranked = None if r.most_common_good_bases is None else [chr(b) for b in r.most_common_good_bases]
            assert ranked == want["ranked"], v["line"]
            for c in want["calls"]:
                p = po.CallerParams(*c["params"])
                base, mask = po.call_record(r, p)
                names = po.filter_names(p)
                failed = [names[i] for i in range(6) if mask >> i & 1] or None
                assert (chr(base), failed) == (c["base"], c["failed"]), (v["line"], c["params"])
                n_calls += 1
    return n_calls


def test_float_threshold_table(pileup_vectors):
    for f, table in pileup_vectors["freq_threshold"].items():
        f = float(f)
        for d, k in enumerate(table):
            assert min(x for x in range(d + 2) if not (x < d * f)) == k


def test_whole_file_runs(pileup_vectors):
    from oracle import fuzz
    from tests.conftest import load_golden
    # the second file holds later additions (deep pileups, long contig names, positions around the powers of ten)
    # ... the third one files with other line ends (CR LF, a mix with lone CRs, '\v' / '\f' before the line end)
    # ... the fourth one files whose contig names are not plain ASCII (text to the reference, UTF-8 bytes to the restatement)
    for run in (pileup_vectors["runs"] + load_golden("pileup_runs2.json.gz")["runs"] + load_golden("pileup_runs3.json.gz")["runs"]
                + load_golden("pileup_runs_utf8.json.gz")["runs"]):
        kw = dict(run["kw"])
        if "contigs" in kw:
            kw["contigs"] = tuple(kw["contigs"])
        data, _, _ = fuzz.synth_pileup(run["seed"], **kw)
        if run.get("line_ends"):
            data = fuzz.with_line_ends(data, run["line_ends"], run["seed"])
        snps = [(c.encode(), p) for c, p in run["snplist"]]
        excl = {(c.encode(), p) for c, p in run["excluded"]}
        cons, _ = po.call_consensus_sites(data, snps, excl, po.CallerParams(*run["params"]))
        assert cons.decode() == run["consensus"], run["seed"]


def test_universal_newlines_and_blank_lines():
    data = b"c 1 A 1 . I\r\nc 2 A 1 . I\rc 3 A 1 . I\nc 4 A 1 . I"
    assert [ln for _, ln in po.iter_lines(data)] == [b"c 1 A 1 . I", b"c 2 A 1 . I", b"c 3 A 1 . I", b"c 4 A 1 . I"]
    with pytest.raises(ValueError):
        list(po.scan_sites(b"c 1 A 1 . I\n\nc 2 A 1 . I\n", {(b"c", 2)}, 0))


def test_steps_vectors(steps_vectors):
    for v in steps_vectors["find_dense_regions"]:
        assert [list(t) for t in so.find_dense_regions(v["m"], v["w"], v["snps"])] == v["out"]
    for v in steps_vectors["merge_regions"]:
        assert [list(t) for t in so.merge_regions([tuple(r) for r in v["in"]])] == v["out"]
    for v in steps_vectors["in_region"]:
        assert so.in_region(v["pos"], [tuple(r) for r in v["regions"]]) == v["out"]
    for v in steps_vectors["sequence_distance"]:
        assert so.sequence_distance(v["a"], v["b"]) == v["d"]
    for v in steps_vectors["collect_all"]:
        samples = [("s%d" % i, [tuple(r) for r in recs]) for i, recs in enumerate(v["samples"])]
        got = so.bad_regions(samples, v["lens"], v["edge"], v["max_snps"], v["windows"], mode="all")
        assert {c: [list(r) for r in regs] for c, regs in got.items()} == v["out"]
    w = steps_vectors["snplist_writer"]
    merged = sorted(((c, p), names) for c, p, names in w["in"])
    assert so.snplist_text(merged) == w["out"]


def _vcf_sites(path):
    out = []
    with open(path) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            c, p = line.split("\t")[:2]
            out.append((c, int(p)))
    return out


@pytest.mark.parametrize("ds", ["lambdaVirus", "agona", "listeria"])
def test_bundled_fixtures(fixture_trees, ds):
    root, meta = fixture_trees[ds]
    sdirs = sorted(os.listdir(os.path.join(root, "samples")))
    samples = [(os.path.join(root, "samples", s), s, _vcf_sites(os.path.join(root, "samples", s, "var.flt.vcf"))) for s in sdirs]
    merged, _ = so.merge_sites(samples)
    assert so.snplist_text(merged) == open(os.path.join(root, "snplist.txt")).read()

    bad = so.bad_regions([(n, recs) for _, n, recs in samples], meta["contig_lengths"], 500, [3, 2, 1], [1000, 125, 15])
    pres = [(d, n, [k for k in recs if not so.in_region(k[1], bad[k[0]])]) for d, n, recs in samples]
    merged_p, _ = so.merge_sites(pres)
    assert so.snplist_text(merged_p) == open(os.path.join(root, "snplist_preserved.txt")).read()

    for suffix in ("", "_preserved"):
        seqs = so.parse_snpma(open(os.path.join(root, "snpma%s.fasta" % suffix)).read())
        ids, d = so.distance_tables(seqs)
        assert so.matrix_text(ids, d) == open(os.path.join(root, "snp_distance_matrix%s.tsv" % suffix)).read()
        pw = os.path.join(root, "snp_distance_pairwise%s.tsv" % suffix)
        if os.path.isfile(pw):
            assert so.pairwise_text(ids, d) == open(pw).read()


def test_vcf_row_known_answers():
    """The reference's own doctest answers for consensus.vcf rows (vcf_writer.py:400-429)."""
    from oracle import vcf_oracle as vo
    r = po.parse_record([b"ID", b"42", b"G", b"0", b"", b""], 15)
    assert vo.vcf_row(r, ["Fail"], ".").split("\t") == ["ID", "42", ".", "G", ".", ".", "Fail", "NS=1", vo.FORMAT_IDS, ".:0:0:0:0:0:0:0:Fail"]
    r = po.parse_record([b"ID", b"42", b"G", b"14", b"aaaaAAAA...,,,", b"00001111222333"], 15)
    assert vo.vcf_row(r, None, ".").split("\t") == ["ID", "42", ".", "G", "A", ".", "PASS", "NS=1", vo.FORMAT_IDS, "1:14:6:8:3:3:4:4:PASS"]
    r = po.parse_record([b"ID", b"42", b"G", b"23", b"TttaaAAAcCC.......,,,,,", b"00011111222333333333333"], 15)
    assert vo.vcf_row(r, None, ".").split("\t") == ["ID", "42", ".", "G", "A,C,T", ".", "PASS", "NS=1", vo.FORMAT_IDS,
                                                    "0:23:12:5,3,3:7:5:3,2,1:2,1,2:PASS"]


def test_depth_sum_against_the_reference_collect_metrics():
    """avePileupDepth as the reference's own collect_metrics() printed it for pileups with every line shape (1-base
    reference, so the printed average is the sum): pins oracle.depth_sum, which the GPU by-product is checked against."""
    from tests.conftest import load_golden
    from oracle import fuzz
    vec = load_golden("metrics_vectors.json.gz")["depth_sum"]
    assert len(vec) >= 6
    for v in vec:
        kw = dict(v["kw"])
        if v["kind"] == "synth":
            if "contigs" in kw:
                kw["contigs"] = tuple(kw["contigs"])
            data = fuzz.synth_pileup(**kw)[0]
        else:
            data = fuzz.odd_depth_lines(kw["seed"], kw["eol"].encode())
        assert len(data) == v["bytes"]
        assert "%.2f" % float(po.depth_sum(data)) == v["avePileupDepth"]


# ---- phase-1 site calling (VarScan mpileup2snp): the restatement against the reference's bundled var.flt.vcf files --------
def _var_flt_vcfs():
    import tarfile
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fixtures")
    for ds in ("lambdaVirus", "agona", "listeria"):
        with tarfile.open(os.path.join(here, ds, "expected.tar.xz")) as t:
            for m in t.getmembers():
                if m.name.endswith("/var.flt.vcf"):
                    yield ds + "/" + m.name, t.extractfile(m).read().decode()


def test_varscan_oracle_reproduces_every_bundled_var_flt_vcf_line():
    """58 files, 69 019 data lines: header text, column layout, PVAL / GQ (Fisher against the 0.1 % error model), FREQ text,
    GT / HOM / HET, ADP — each line rebuilt from its own counts.  And the selection rules hold on every line."""
    from oracle import varscan_oracle as vo
    prm = vo.Params(**vo.PIPELINE_DEFAULTS)
    n_files = n_rows = one_sided = 0
    for name, text in _var_flt_vcfs():
        n_files += 1
        lines = text.splitlines(True)
        assert "".join(ln for ln in lines if ln.startswith("#")) == vo.VCF_HEADER % {"q": 15}, name
        for ln in lines:
            if ln.startswith("#"):
                continue
            chrom, pos, r = vo.row_from_fixture_line(ln)
            assert vo.vcf_row(chrom, pos, r) == ln, (name, ln)
            n_rows += 1
            # the selection tests, from inside
            assert r["SDP"] >= prm.min_coverage and r["DP"] >= prm.min_coverage and r["AD"] >= prm.min_reads2 and r["ABQ"] >= prm.min_avg_qual
            assert float(r["AD"]) / float(r["total"]) >= prm.min_var_freq and r["p"] <= prm.p_value and r["hom"]
            assert r["total"] >= r["RD"] + r["AD"] and r["RDF"] + r["RDR"] == r["RD"] and r["ADF"] + r["ADR"] == r["AD"]
            # the strand filter as restated passes every bundled line (all are PASS)
            var_plus = r["ADF"] / float(r["AD"])
            if var_plus < 0.1 or var_plus > 0.9:
                one_sided += 1
                if r["RD"] > 1:
                    ref_plus = r["RDF"] / float(r["RD"])
                    assert not (vo.two_tailed_p(r["RDF"], r["RDR"], r["ADF"], r["ADR"]) < 0.01 and 0.1 <= ref_plus <= 0.9), ln
    assert (n_files, n_rows) == (58, 69019) and one_sided > 1000


def test_varscan_oracle_read_counting_rules():
    """The read-base walk as restated (UNPINNED by reference data — see the module header): which bytes own a quality,
    what counts where."""
    from oracle import varscan_oracle as vo
    c = vo.read_counts(b".,Aa^]G$*Nn+2AC.-1g,", b"IIIIIIIIIII", 15)
    # . , A a (^] skipped) G ($ skipped) * N n (+2AC: one indel read) . (-1g: one indel read) ,
    assert c.ref == [2, 2, 4 * 40] and c.alt == {"A": [1, 1, 80], "G": [1, 0, 40]} and c.indel == 2 and c.total() == 9
    assert vo.quality_depth(b"IIIIIIIIIII", 15) == 11
    # qualities below the threshold do not count, but still advance the cursor; a short quality string reads as quality 0
    c = vo.read_counts(b"AAAA", b"I#I", 15)
    assert c.alt == {"A": [2, 0, 80]}
    # three-digit and four-digit indel lengths, an indel that runs off the end, a sign without digits
    assert vo.read_counts(b".+12ACGTACGTACGT.", b"II", 15).ref == [2, 0, 80]
    assert vo.read_counts(b".+3AC", b"I", 15).indel == 1 and vo.read_counts(b".+A", b"II", 15).alt == {"A": [1, 0, 40]}
    # one line end to end: 9 of 10 reads G at quality 40 over an A reference
    r = vo.call_line("a", 10, b"GGGGggggg.", b"I" * 10, vo.Params(**vo.PIPELINE_DEFAULTS))
    assert vo.vcf_row("c", "7", r) == "c\t7\t.\tA\tG\t.\tPASS\tADP=10;WT=0;HET=0;HOM=1;NC=0\tGT:GQ:SDP:DP:RD:AD:FREQ:PVAL:RBQ:ABQ:RDF:RDR:ADF:ADR\t" \
                                       "1/1:42:10:10:1:9:90%:5.9538E-5:40:40:1:0:4:5\n"       # (GQ / PVAL of RD 1, AD 9 as in agona ERR178926 pos 226973)


def test_files_with_one_odd_line_end_as_the_reference_does():
    """badline_runs.json.gz: the reference's own driver on files with one odd line — the exception class it raises, or the
    consensus when the line is none of its business (pileup.py:423-429 only looks at chrom and position of unlisted lines)."""
    from oracle import fuzz
    from tests.conftest import load_golden
    runs = load_golden("badline_runs.json.gz")["runs"]
    assert {r["scenario"] for r in runs} == set(fuzz.BAD_LINE_SCENARIOS)
    for run in runs:
        kw = dict(run["kw"])
        if "contigs" in kw:
            kw["contigs"] = tuple(kw["contigs"])
        base, _, _ = fuzz.synth_pileup(run["seed"], **kw)
        snps = [(c.encode(), p) for c, p in run["snplist"]]
        data = fuzz.with_bad_line(base, run["scenario"], set(snps))
        if "exception" in run:
            with pytest.raises((ValueError, IndexError)) as ei:
                po.call_consensus_sites(data, snps, set(), po.CallerParams(*run["params"]))
            assert type(ei.value).__name__ == run["exception"], run["scenario"]
        else:
            cons, _ = po.call_consensus_sites(data, snps, set(), po.CallerParams(*run["params"]))
            assert cons.decode() == run["consensus"], run["scenario"]


def test_filter_regions_runs_of_the_reference_driver():
    """filter_runs.json.gz: which records the reference's own filter_regions driver preserves and removes per sample (mode
    all / each, outgroup samples, two rule sets; the VCF writer replaced by a stand-in that records CHROM and POS)."""
    from oracle import fuzz
    from tests.conftest import load_golden
    runs = load_golden("filter_runs.json.gz")["runs"]
    assert {(r["mode"], bool(r["outgroup"])) for r in runs} == {("all", False), ("all", True), ("each", False), ("each", True)}
    for run in runs:
        lengths, cohort = fuzz.vcf_cohort(run["seed"])
        samples = sorted(cohort.items())
        bad = so.bad_regions(samples, lengths, run["edge"], run["max_snps"], run["windows"], mode=run["mode"], outgroup=set(run["outgroup"]))
        for name, recs in samples:
            want = run["result"][name]
            if name in run["outgroup"]:
                got_p, got_r = [list(r) for r in recs], []
            else:
                regions = bad if run["mode"] == "all" else bad[name]
                got_r = [list(r) for r in recs if so.in_region(r[1], regions.get(r[0], []))]
                got_p = [list(r) for r in recs if not so.in_region(r[1], regions.get(r[0], []))]
            assert (got_p, got_r) == (want["preserved"], want["removed"]), (run["mode"], run["outgroup"], run["edge"], name)


def test_merge_sites_runs_of_the_reference_driver():
    """merge_runs.json.gz: snplist text and filtered sample list of the reference's own merge_sites driver, with --maxsnps
    limits that take samples out; VCFs that repeat a position."""
    from oracle import fuzz
    from tests.conftest import load_golden
    for run in load_golden("merge_runs.json.gz")["runs"]:
        _, cohort = fuzz.vcf_cohort(run["seed"])
        samples = [("$W/" + name, name, recs + recs[:2]) for name, recs in sorted(cohort.items())]
        merged, excluded = so.merge_sites(samples, run["max_snps"])
        assert so.snplist_text(merged) == run["snplist"], (run["seed"], run["max_snps"])
        listed = "".join(d + "\n" for d, _, _ in reversed(samples) if d not in excluded)
        assert listed == run["filtered"], (run["seed"], run["max_snps"])


def test_distance_runs_of_the_reference_driver():
    """distance_runs.json.gz: both TSV texts (or the exception class) of the reference's own distance driver on the untidy SNP
    matrix files of fuzz.untidy_snpmas: ids out of order, a repeated id, growing lengths, a shorter later sequence, CR LF, an
    empty record, text before the first header."""
    from oracle import fuzz
    from tests.conftest import load_golden
    texts = dict(fuzz.untidy_snpmas())
    for run in load_golden("distance_runs.json.gz")["runs"]:
        text = texts[run["name"]].replace("\r\n", "\n")
        if "exception" in run:
            with pytest.raises(Exception) as ei:
                so.distance_tables(so.parse_snpma(text))
            assert type(ei.value).__name__ == run["exception"], run["name"]
            continue
        ids, table = so.distance_tables(so.parse_snpma(text))
        assert so.pairwise_text(ids, table) == run["pairwise"], run["name"]
        assert so.matrix_text(ids, table) == run["matrix"], run["name"]

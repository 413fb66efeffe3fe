"""Host-side mirror of the reference's CLI / conventions, checked without a GPU."""
import os
import sys
import time

import pytest

from tests.conftest import load_golden


def test_cli_parser_matches_reference_parser():
    """argparse Namespaces equal those of the real reference parser (vectors from oracle/gen_golden.py)."""
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    for v in load_golden("cli_vectors.json.gz"):
        ns = vars(cli.parse_command_line(v["line"]))
        # (the two --amdMetrics* options are extensions of this build: additive, absent from the reference's Namespace)
        got = {k: val for k, val in ns.items() if k not in ("func", "excepthook") and not k.startswith("amd")}
        got["excepthook"] = ns["excepthook"].__name__
        got["func"] = ns["func"].__name__
        assert got == v["args"], v["line"]


def test_cli_argument_errors_exit_2(capsys):
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    for line in ("call_consensus -c 0.5 x.pileup", "call_consensus -b 0.6 x.pileup", "distance", "filter_regions only_one_arg",
                 "call_consensus --vcfFailedSnpGt 2 x.pileup"):
        with pytest.raises(SystemExit) as ei:
            cli.parse_command_line(line)
        assert ei.value.code == 2
        assert capsys.readouterr().err.startswith("Error: ")


def test_filter_regions_list_validation_exits_100(tmp_path, monkeypatch):
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    log = tmp_path / "error.log"
    monkeypatch.setenv("errorOutputFile", str(log))
    for line, msg in (("filter_regions d r -w 1000 100 -m 3", "same number of arguments"),
                      ("filter_regions d r -w 0 -m 3", "length of the window must be a positive integer"),
                      ("filter_regions d r -w 10 -m 0", "maximum number of SNPs allowed must be a positive integer"),
                      ("filter_regions d r -l 0", "length of the edge regions must be a positive integer")):
        with pytest.raises(SystemExit) as ei:
            cli.parse_command_line(line)
        assert ei.value.code == 100
        assert msg in log.read_text()


def test_error_protocol(tmp_path, monkeypatch, capsys):
    from snp_pipeline_amd import utils
    log = tmp_path / "error.log"
    monkeypatch.setenv("errorOutputFile", str(log))
    monkeypatch.setattr(sys, "argv", ["cfsan_snp_pipeline", "call_consensus", "x"])
    with pytest.raises(SystemExit) as ei:
        utils.global_error("Error: boom.")
    assert ei.value.code == 100
    assert log.read_text() == "cfsan_snp_pipeline call_consensus failed.\nError: boom.\n" + "=" * 80 + "\n"
    log.write_text("")
    monkeypatch.delenv("StopOnSampleError", raising=False)
    with pytest.raises(SystemExit) as ei:
        utils.sample_error("Error: sample.", continue_possible=True)
    assert ei.value.code == 100                               # unset -> stop
    monkeypatch.setenv("StopOnSampleError", "false")
    log.write_text("")
    utils.sample_error("Error: sample.", continue_possible=True)          # continues
    assert log.read_text() == "cfsan_snp_pipeline call_consensus\nError: sample.\n" + "=" * 80 + "\n"
    with pytest.raises(SystemExit) as ei:
        utils.sample_error("Error: sample.", continue_possible=False)
    assert ei.value.code == 98
    try:
        raise ValueError("bad line")
    except ValueError:
        with pytest.raises(SystemExit) as ei:
            utils.handle_sample_exception(*sys.exc_info())
        assert ei.value.code == 98
        with pytest.raises(SystemExit) as ei:
            utils.handle_global_exception(*sys.exc_info())
        assert ei.value.code == 100
    assert "ValueError exception in function test_error_protocol" in log.read_text()
    capsys.readouterr()


def test_target_needs_rebuild(tmp_path):
    from snp_pipeline_amd import utils
    src, tgt = tmp_path / "src", tmp_path / "tgt"
    src.write_text("x")
    assert utils.target_needs_rebuild([str(src)], str(tgt))               # missing
    tgt.write_text("")
    assert utils.target_needs_rebuild([str(src)], str(tgt))               # empty
    tgt.write_text("y")
    now = time.time()
    os.utime(str(src), (now - 10, now - 10))
    os.utime(str(tgt), (now, now))
    assert not utils.target_needs_rebuild([str(src), str(tmp_path / "absent")], str(tgt))
    os.utime(str(src), (now + 10, now + 10))
    assert utils.target_needs_rebuild([str(src)], str(tgt))


def test_text_codecs(tmp_path, steps_vectors):
    from snp_pipeline_amd import utils
    w = steps_vectors["snplist_writer"]
    rows = sorted(((c, p), names) for c, p, names in w["in"])
    path = tmp_path / "snplist.txt"
    utils.write_list_of_snps(str(path), [k for k, _ in rows], [n for _, n in rows])
    assert path.read_text() == w["out"]
    assert [list(t) for t in utils.read_snp_position_list(str(path))] == steps_vectors["snplist_reader"]
    (tmp_path / "bad.txt").write_text("chr\tnotanumber\n")
    with pytest.raises(ValueError):
        utils.read_snp_position_list(str(tmp_path / "bad.txt"))
    fa = tmp_path / "r.fasta"
    fa.write_text(">c1 description here\nACGT\nAC\n>c2\n\nGG\n")
    assert utils.read_fasta_lengths(str(fa)) == {"c1": 6, "c2": 2}
    out = tmp_path / "o.fasta"
    with open(str(out), "w") as f:
        utils.write_fasta_record(f, "s1", "A" * 130)
        utils.write_fasta_record(f, "s2", "")
    assert out.read_text() == ">s1\n" + "A" * 60 + "\n" + "A" * 60 + "\n" + "A" * 10 + "\n>s2\n"


def test_vcf_header_reorder_matches_lambda_fixture(fixture_trees):
    from snp_pipeline_amd import filter_regions as fr
    from snp_pipeline_amd import utils
    root, _ = fixture_trees["lambdaVirus"]
    for s in sorted(os.listdir(os.path.join(root, "samples"))):
        d = os.path.join(root, "samples", s)
        header, data, sites = utils.read_vcf_sites(os.path.join(d, "var.flt.vcf"))
        want = [ln for ln in open(os.path.join(d, "var.flt_preserved.vcf")) if ln.startswith("#")]
        assert fr.reorder_header(header) == want
        assert len(data) == len(sites) > 0


def test_consensus_vcf_header_matches_lambda_fixture(fixture_trees):
    from snp_pipeline_amd import vcf_writer
    root, _ = fixture_trees["lambdaVirus"]
    want = [ln.rstrip("\n") for ln in open(os.path.join(root, "samples", "sample1", "consensus.vcf")) if ln.startswith("#")]
    filters = vcf_writer.filter_descriptions(0.6, 3, 0, 0.0)
    got = vcf_writer.header_lines("sample1", filters, "lambda_virus.fasta")
    skip = ("##fileDate", "##source")                           # the reference's test ignores these two (test_cfsan_snp_pipeline.py:173)
    assert [x for x in got if not x.startswith(skip)] == [x for x in want if not x.startswith(skip)]


def test_snp_reference_matches_lambda_fixture(tmp_path, fixture_trees):
    """referenceSNP.fasta / referenceSNP_preserved.fasta of the bundled lambda results, byte for byte, through the CLI
    (snp_reference.py:12-77; the other two datasets ship no reference fasta)."""
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    from snp_pipeline_amd import snp_reference
    tree, _ = fixture_trees["lambdaVirus"]
    ref = os.path.join(os.path.dirname(__file__), "golden", "fixtures", "lambdaVirus", "lambda_virus.fasta")
    for snplist, want in (("snplist.txt", "referenceSNP.fasta"), ("snplist_preserved.txt", "referenceSNP_preserved.fasta")):
        out = str(tmp_path / want)
        rc = cli.run_command_from_args(cli.parse_argument_list(
            ["snp_reference", "-v", "0", "-l", os.path.join(tree, snplist), "-o", out, ref]))
        assert rc == 0
        assert open(out, "rb").read() == open(os.path.join(tree, want), "rb").read()
    # python indexing of the reference: position 0 reads the last base, a position past the end raises
    fa = tmp_path / "r.fasta"
    fa.write_text(">c2 desc\nacgt\nTT\n>c1\nGGa\n")
    sl = tmp_path / "s.txt"
    sl.write_text("c2\t1\t1\tx\nc1\t3\t1\tx\nc2\t0\t1\tx\nzz\t9\t1\tx\nc2 6 1 x\n")
    snp_reference.write_reference_snp_file(str(fa), str(sl), str(tmp_path / "o.fasta"))
    assert (tmp_path / "o.fasta").read_text() == ">c1\nA\n>c2\nATT\n"
    sl.write_text("c1\t4\t1\tx\n")
    with pytest.raises(IndexError):
        snp_reference.write_reference_snp_file(str(fa), str(sl), str(tmp_path / "o.fasta"))


def _counts_from_vcf_row(row):
    """Invert one consensus.vcf data line into (chrom, pos, snpgpu_site_counts record, ranked symbols)."""
    import numpy as np
    from snp_pipeline_amd import _lib as L
    from snp_pipeline_amd.device import COUNTS_DTYPE
    f = row.split("\t")
    chrom, pos, ref, alt_s, ft = f[0], int(f[1]), f[3], f[4], f[6]
    assert f[2] == "." and f[5] == "." and f[7] == "NS=1" and f[8] == "GT:SDP:RD:AD:RDF:RDR:ADF:ADR:FT"
    gt, sdp, rd, ad, rdf, rdr, adf, adr, ft2 = f[9].split(":")
    assert ft2 == ft
    alts = [] if alt_s == "." else alt_s.split(",")
    tot = {a: int(x) for a, x in zip(alts, ad.split(","))} if alts else {}
    fwd = {a: int(x) for a, x in zip(alts, adf.split(","))} if alts else {}
    rev = {a: int(x) for a, x in zip(alts, adr.split(","))} if alts else {}
    if not alts:
        assert (ad, adf, adr) == ("0", "0", "0")
    if int(rd) > 0:
        tot[ref], fwd[ref], rev[ref] = int(rd), int(rdf), int(rdr)
    ranked = sorted(tot, key=lambda s: (-tot[s], s))                 # pileup.py:263-266: count descending, byte ascending
    assert [s for s in ranked if s != ref] == alts                   # ... which is the order the reference printed the ALTs in
    c = np.zeros(1, dtype=COUNTS_DTYPE)[0]
    c["raw_depth"], c["good_depth"] = int(sdp), sum(tot.values())
    c["fwd_good_depth"], c["rev_good_depth"] = sum(fwd.values()), sum(rev.values())
    c["n_symbols"], c["ref_base"], c["status"] = len(ranked), ord(ref), L.ST_OK
    for r, s in enumerate(ranked):
        c["sym"][r], c["total"][r], c["fwd"][r], c["rev"][r] = ord(s), tot[s], fwd[s], rev[s]
    return chrom, pos, c, ranked, ft, gt


def test_consensus_vcf_rows_and_fasta_of_the_lambda_fixtures(fixture_trees):
    """Every data line of the eight bundled lambda consensus*.vcf files (the reference's own output on real reads): the host
    writer and the oracle's row function reproduce it byte for byte from the counts it encodes, the GT / FT rules included,
    and (counts -> FASTA character) agrees with the bundled consensus*.fasta (call_consensus.py:161-188)."""
    from oracle import pileup_oracle as po
    from oracle import vcf_oracle as vo
    from snp_pipeline_amd import utils, vcf_writer
    root, _ = fixture_trees["lambdaVirus"]
    n_rows = 0
    for suffix in ("", "_preserved"):
        snplist = utils.read_snp_position_list(os.path.join(root, "snplist%s.txt" % suffix))
        for s in sorted(os.listdir(os.path.join(root, "samples"))):
            sdir = os.path.join(root, "samples", s)
            lines = open(os.path.join(sdir, "consensus%s.vcf" % suffix)).read().split("\n")
            names = [ln.split("ID=")[1].split(",")[0] for ln in lines if ln.startswith("##FILTER=") and "ID=PASS" not in ln]
            assert names == ["RawDpth", "VarFreq60", "Depth3", "StrDpth0", "StrBias0", "Region"]
            fasta = "".join(open(os.path.join(sdir, "consensus%s.fasta" % suffix)).read().split("\n")[1:])
            assert len(fasta) == len(snplist)
            called = {}
            recs, rec_keys, rec_rows = [], [], []
            for row in lines:
                if not row or row.startswith("#"):
                    continue
                chrom, pos, c, ranked, ft, gt = _counts_from_vcf_row(row)
                recs.append(c), rec_keys.append(pos), rec_rows.append(row)
                mask = 0
                for name in ([] if ft == "PASS" else ft.split(";")):
                    mask |= 1 << names.index(name)
                c["filters"] = mask
                recs[-1] = c
                assert vcf_writer.row_from_counts(chrom, pos, c, names, False, ".") == row
                rec = po.Record(chrom.encode(), pos, bytes([int(c["ref_base"])]), int(c["raw_depth"]), int(c["good_depth"]),
                                int(c["fwd_good_depth"]), int(c["rev_good_depth"]),
                                {int(c["sym"][r]): int(c["total"][r]) for r in range(len(ranked))},
                                {int(c["sym"][r]): int(c["fwd"][r]) for r in range(len(ranked))},
                                {int(c["sym"][r]): int(c["rev"][r]) for r in range(len(ranked))},
                                [ord(x) for x in ranked] if ranked else None)
                assert vo.vcf_row(rec, None if ft == "PASS" else ft.split(";"), ".") == row
                # GT: '.' when a filter failed or nothing was counted, else 0 / 1 by whether the top symbol is the reference
                assert gt == ("." if (mask or not ranked) else ("0" if ranked[0] == chr(int(c["ref_base"])) else "1"))
                called[(chrom, pos)] = "-" if (mask or not ranked or ranked[0] == "*") else ranked[0]
                n_rows += 1
            # ... and from the library's formatter (what the CLI writes), all rows of the file at once
            import numpy as np
            cname = np.frombuffer(chrom.encode(), dtype=np.uint8)
            text = vcf_writer.format_rows(np.array(recs), np.arange(len(recs)), cname, np.array([0, len(cname)], dtype=np.uint32),
                                          np.array(rec_keys, dtype=np.uint64), names, False, ".")
            assert text.decode() == "".join(r + "\n" for r in rec_rows)
            want = "".join(called.get(key, "-") for key in snplist)
            assert want == fasta, (s, suffix)
            # consensus.vcf holds the snplist positions that have a pileup line; the preserved run adds the excluded ones
            if not suffix:
                assert set(called) <= set(snplist)
    assert n_rows > 1200


def test_snp_reference_matches_listeria_fixture(tmp_path, fixture_trees):
    """The second data set that ships its reference: referenceSNP*.fasta of the listeria results (10 102 / 1 040 sites)."""
    import lzma
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    tree, _ = fixture_trees["listeria"]
    src = os.path.join(os.path.dirname(__file__), "golden", "fixtures", "listeria", "CFSAN023463.HGAP.draft.fasta.xz")
    ref = tmp_path / "CFSAN023463.HGAP.draft.fasta"
    ref.write_bytes(lzma.decompress(open(src, "rb").read()))
    for snplist, want in (("snplist.txt", "referenceSNP.fasta"), ("snplist_preserved.txt", "referenceSNP_preserved.fasta")):
        out = str(tmp_path / want)
        assert cli.run_command_from_args(cli.parse_argument_list(
            ["snp_reference", "-v", "0", "-l", os.path.join(tree, snplist), "-o", out, str(ref)])) == 0
        assert open(out, "rb").read() == open(os.path.join(tree, want), "rb").read()


def test_vcf_reader_takes_the_first_line_as_header_when_there_is_no_chrom_line(tmp_path):
    """PyVCF's Reader takes the first line that does not start with "##" as the column header, whatever it says, and the Writer
    prints it back as '#' + the TAB-joined columns minus the first byte: the reference's regression tests feed "Dummy vcf
    content" as var.flt.vcf and expect filter_regions / merge_sites to reach their output files (regression_tests.sh:2030-2093)."""
    from snp_pipeline_amd import utils
    dummy = tmp_path / "dummy.vcf"
    dummy.write_text("Dummy vcf content\n")
    assert utils.read_vcf_sites(str(dummy)) == (["#ummy\tvcf\tcontent\n"], [], [])
    assert utils.read_vcf_site_arrays(str(dummy))[2].size == 0
    odd = tmp_path / "odd.vcf"
    odd.write_text("##fileformat=VCFv4.1\nthis is  not a header\nchr\t12\t.\n")
    assert utils.read_vcf_sites(str(odd)) == (["##fileformat=VCFv4.1\n", "#his\tis\tnot\ta\theader\n"], ["chr\t12\t.\n"], [("chr", 12)])
    ok = tmp_path / "ok.vcf"
    ok.write_text("##fileformat=VCFv4.1\n#CHROM\tPOS\n\nc1\t5\t.\nc1\t9\t.\n")
    assert utils.read_vcf_sites(str(ok))[2] == [("c1", 5), ("c1", 9)]


def test_metrics_properties_update_and_missing_positions(tmp_path, fixture_trees):
    """The metrics side file: name=value lines updated in place; missingPos is the number of '-' of the sample's consensus
    record (collect_metrics.py:109-128) — checked against the bundled lambda metrics files."""
    from snp_pipeline_amd import utils
    path = tmp_path / "metrics"
    utils.update_properties(str(path), {"avePileupDepth": "23.10", "missingPos": "4"})
    assert path.read_text() == "avePileupDepth=23.10\nmissingPos=4\n"
    path.write_text('sample="s1"\n# note\nmissingPos=9\nmachine=\n')
    utils.update_properties(str(path), {"missingPos": "0", "avePileupDepth": "1.00"})
    assert path.read_text() == 'sample="s1"\n# note\nmissingPos=0\nmachine=\navePileupDepth=1.00\n'
    root, _ = fixture_trees["lambdaVirus"]
    checked = 0
    for s in sorted(os.listdir(os.path.join(root, "samples"))):
        sdir = os.path.join(root, "samples", s)
        props = dict(ln.rstrip("\n").split("=", 1) for ln in open(os.path.join(sdir, "metrics")) if "=" in ln)
        for fasta, key in (("consensus.fasta", "missingPos"), ("consensus_preserved.fasta", "missingPosPreserved")):
            seq = "".join(open(os.path.join(sdir, fasta)).read().split("\n")[1:])
            assert str(seq.count("-")) == props[key].strip('"'), (s, key)
            checked += 1
    assert checked == 8


def test_device_slots_spread_processes(tmp_path):
    """acquire_device_slot: what the array of per-sample CLI processes (run.py:709-710) uses to pick a GPU — slot 0 of every
    device before slot 1 of any, never more than max_per_device holders per device, the lock gone with its holder."""
    import multiprocessing as mp
    from snp_pipeline_amd import device as dev
    lock_dir = str(tmp_path / "locks")
    held = [dev.acquire_device_slot(3, max_per_device=2, lock_dir=lock_dir) for _ in range(6)]
    per_dev = {}
    for d_, _ in held:
        per_dev[d_] = per_dev.get(d_, 0) + 1
    assert per_dev == {0: 2, 1: 2, 2: 2}
    assert sorted(d_ for d_, _ in held[:3]) == [0, 1, 2]            # the first three processes land on three different GPUs
    assert sorted(os.listdir(lock_dir)) == ["dev%d.slot%d" % (d_, j) for d_ in range(3) for j in range(2)]
    # everything is taken: a seventh process waits, and gets the slot the moment a holder goes away
    q = mp.get_context("spawn").Queue()
    p = mp.get_context("spawn").Process(target=_slot_worker, args=(lock_dir, q))
    p.start()
    time.sleep(1.0)
    assert q.empty()
    gone = held.pop(2)
    gone[1].close()
    got = q.get(timeout=30)
    p.join(30)
    assert 0 <= got < 3
    for _, f in held:
        f.close()


def _slot_worker(lock_dir, q):
    from snp_pipeline_amd import device as dev
    d_, f = dev.acquire_device_slot(3, max_per_device=2, lock_dir=lock_dir)
    q.put(d_)
    f.close()


def test_call_consensus_batch_parser_and_all_positions_writer(tmp_path):
    """The extension subcommand takes call_consensus's options; the --vcfAllPos writer lays out one row per line."""
    import numpy as np
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    from snp_pipeline_amd import vcf_writer
    ns = cli.parse_command_line("call_consensus_batch -l snplist.txt -o consensus_preserved.fasta -e var.flt_removed.vcf -q 15 -c 0.9 "
                                "--vcfFileName consensus_preserved.vcf --pileupName reads.all.pileup dirs.txt")
    assert (ns.sampleDirsFile, ns.consensusFile, ns.excludeFile, ns.minBaseQual, ns.minConsFreq, ns.pileupName) == \
        ("dirs.txt", "consensus_preserved.fasta", "var.flt_removed.vcf", 15, 0.9, "reads.all.pileup")
    assert ns.func.__name__ == "call_consensus_batch" and ns.vcfAllPos is False
    one = cli.parse_command_line("call_consensus --amdMetricsRefFasta ref.fa x.pileup")
    assert one.amdMetricsRefFasta == "ref.fa" and one.amdMetricsFile is None
    pile = tmp_path / "p.pileup"
    pile.write_bytes(b"c1\t5\tG\t2\t.A\tII\nlong_contig_name_%s\t77\ta\t0\t*\t*\n" % (b"x" * 300))
    rows = [_counts_from_vcf_row("c1\t5\t.\tG\tA\t.\tPASS\tNS=1\tGT:SDP:RD:AD:RDF:RDR:ADF:ADR:FT\t1:2:1:1:1:0:1:0:PASS")[2],
            _counts_from_vcf_row("z\t77\t.\tA\t.\t.\tRawDpth\tNS=1\tGT:SDP:RD:AD:RDF:RDR:ADF:ADR:FT\t.:0:0:0:0:0:0:0:RawDpth")[2]]
    rows[1]["filters"] = 1
    rows[1]["ref_base"] = ord("a")
    args = cli.parse_command_line("call_consensus --vcfFileName v.vcf --vcfAllPos --vcfRefName r.fa x.pileup")
    counts = np.array(rows)
    vcf_writer.write_all_positions_vcf(str(tmp_path / "v.vcf"), "s1", args, str(pile), np.array([1, 1 + len(b"c1\t5\tG\t2\t.A\tII\n")], dtype=np.uint64), counts)
    data = [x for x in (tmp_path / "v.vcf").read_text().split("\n") if x and not x.startswith("#")]
    assert data[0] == "c1\t5\t.\tG\tA\t.\tPASS\tNS=1\tGT:SDP:RD:AD:RDF:RDR:ADF:ADR:FT\t1:2:1:1:1:0:1:0:PASS"   # a tie ranks A before G: GT 1
    assert data[1].startswith("long_contig_name_" + "x" * 300 + "\t77\t.\tA\t.\t.\tRawDpth\t")


def test_library_vcf_formatter_equals_python_rows():
    """snpgpu_format_vcf_rows against row_from_counts on random records: every GT option, both reference cases, 0 to 8
    ranked symbols, every filter mask, multi-contig keys, a permuted row order."""
    import random
    import numpy as np
    from snp_pipeline_amd import vcf_writer
    from snp_pipeline_amd.device import COUNTS_DTYPE
    rng = random.Random(5)
    names = ["RawDpth", "VarFreq60", "Depth3", "StrDpth0", "StrBias0", "Region"]
    contigs = [b"c", b"gi|9626243|ref|NC_001416.1|", b"NODE_1_length_419034_cov_23.1"]
    cname = np.frombuffer(b"".join(contigs), dtype=np.uint8)
    coff = np.cumsum([0] + [len(c) for c in contigs]).astype(np.uint32)
    n = 400
    recs = np.zeros(n, dtype=COUNTS_DTYPE)
    keys = np.zeros(n, dtype=np.uint64)
    for i in range(n):
        c = recs[i]
        k = rng.choice([0, 0, 1, 1, 2, 3, 6, 8])
        syms = rng.sample(list(b"*ACGTN#<"), k)
        tot = sorted((rng.randint(1, 300) for _ in range(k)), reverse=True)
        c["ref_base"] = rng.choice(list(b"ACGTNacgtn"))
        c["raw_depth"] = rng.randint(0, 100000)
        c["good_depth"] = sum(tot)
        c["n_symbols"] = k
        c["filters"] = rng.randint(0, 63) if rng.random() < 0.5 else 0
        for r in range(k):
            f = rng.randint(0, tot[r])
            c["sym"][r], c["total"][r], c["fwd"][r], c["rev"][r] = syms[r], tot[r], f, tot[r] - f
        keys[i] = (rng.randrange(3) << 32) | rng.randint(0, 4_000_000_000)
    order = np.array(rng.sample(range(n), n), dtype=np.uint32)
    for gt in (".", "0", "1"):
        for keep_case in (False, True):
            text = vcf_writer.format_rows(recs, order, cname, coff, keys, names, keep_case, gt).decode()
            want = "".join(vcf_writer.row_from_counts(contigs[int(keys[j]) >> 32].decode(), int(keys[j]) & 0xFFFFFFFF, recs[j], names, keep_case, gt) + "\n"
                           for j in order)
            assert text == want
    recs[7]["n_symbols"] = 9
    with pytest.raises(ValueError):
        vcf_writer.format_rows(recs, order, cname, coff, keys, names, False, ".")
    # more than eight symbols: ranks 8.. come from the spill record the position's n_symbols points at
    from snp_pipeline_amd.device import SPILL_DTYPE
    spill = np.zeros(3, dtype=SPILL_DTYPE)
    for i, extra in ((7, 1), (20, 8), (33, 120)):
        c = recs[i]
        alphabet = [x for x in range(33, 127) if not (97 <= x <= 122)] + list(range(161, 256))
        syms = rng.sample(alphabet, 8 + extra)
        tot = sorted((rng.randint(1, 300) for _ in syms), reverse=True)
        slot = {7: 0, 20: 1, 33: 2}[i]
        c["n_symbols"] = (8 + extra) | ((slot + 1) << 8)
        c["good_depth"] = sum(tot)
        c["ref_base"] = syms[rng.randrange(len(syms))] if rng.random() < 0.7 else ord("A")
        spill[slot]["n"] = extra
        for r, (sym, t) in enumerate(zip(syms, tot)):
            f = rng.randint(0, t)
            tgt, k = (c, r) if r < 8 else (spill[slot], r - 8)
            tgt["sym"][k], tgt["total"][k], tgt["fwd"][k], tgt["rev"][k] = sym, t, f, t - f
    recs[33]["ref_base"] = ord("A")                             # (a latin-1 symbol as REF would not be ASCII text)
    text = vcf_writer.format_rows(recs, order, cname, coff, keys, names, True, ".", spill=spill)
    want = "".join(vcf_writer.row_from_counts(contigs[int(keys[j]) >> 32].decode(), int(keys[j]) & 0xFFFFFFFF, recs[j], names, True, ".", spill=spill) + "\n"
                   for j in order)
    assert text == want.encode("latin-1")
    assert max(ln.split(b"\t")[4].count(b",") for ln in text.split(b"\n") if ln) >= 126
    with pytest.raises(ValueError):                             # a record that points past the spill it is given
        vcf_writer.format_rows(recs, order, cname, coff, keys, names, True, ".", spill=spill[:2])
    # reference fields of several bytes: up to 64 in the record's ref[], longer ones go on — raw — in the records behind it
    from snp_pipeline_amd.device import spill_reference_field
    size = SPILL_DTYPE.itemsize
    fields = [(50, b"Ac"), (60, bytes(rng.choice(b"ACGTacgtn.,") for _ in range(64))), (70, bytes(rng.choice(b"ACGTacgtn.,") for _ in range(65))),
              (80, bytes(rng.choice(b"ACGTacgt") for _ in range(64 + size))), (90, bytes(rng.choice(b"ACGTacgt") for _ in range(64 + size + 1)))]
    spill2 = np.zeros(3 + sum(1 + (max(len(f) - 64, 0) + size - 1) // size for _, f in fields), dtype=SPILL_DTYPE)
    spill2[:3] = spill
    raw = spill2.view(np.uint8).reshape(-1)
    slot = 3
    for i, field in fields:
        recs[i]["n_symbols"] = (int(recs[i]["n_symbols"]) & 0xFF) | ((slot + 1) << 8)
        recs[i]["ref_base"] = field[0]
        spill2[slot]["ref_len"] = len(field)
        at = slot * size + size - 64
        raw[at:at + len(field)] = np.frombuffer(field, dtype=np.uint8)
        assert spill_reference_field(spill2, slot) == field
        slot += 1 + (max(len(field) - 64, 0) + size - 1) // size
    assert slot == len(spill2) and int(spill2[-1]["n"]) != 0    # (the last record is the tail of the 1 705-byte field: raw bytes, no header)
    for keep_case in (False, True):
        text = vcf_writer.format_rows(recs, order, cname, coff, keys, names, keep_case, "1", spill=spill2)
        want = "".join(vcf_writer.row_from_counts(contigs[int(keys[j]) >> 32].decode(), int(keys[j]) & 0xFFFFFFFF, recs[j], names, keep_case, "1", spill=spill2) + "\n"
                       for j in order)
        assert text == want.encode("latin-1")
        for i, field in fields:
            row = [ln for ln in text.split(b"\n") if ln][list(order).index(i)].split(b"\t")
            assert row[3] == (field if keep_case else field.upper())
    with pytest.raises(ValueError):                             # the tail of a long field has to be there
        vcf_writer.format_rows(recs, order, cname, coff, keys, names, True, ".", spill=spill2[:-1])
    with pytest.raises(ValueError):
        spill_reference_field(spill2[:-1], len(spill2) - 3)


def test_library_distance_tsv_writer_equals_reference_layout(tmp_path, fixture_trees):
    """csrc/tsv_out.hip against the print loops of distance.py:100-114 written out in Python, and against the bundled
    snp_distance_*.tsv of the lambda fixture (matrix values parsed back from the pairwise file)."""
    import numpy as np
    from snp_pipeline_amd import distance

    def py_pairwise(path, ids, mat):
        with open(path, "w") as out:
            out.write("%s\n" % "\t".join(["Seq1", "Seq2", "Distance"]))
            for i, id1 in enumerate(ids):
                for j, id2 in enumerate(ids):
                    out.write("%s\t%s\t%i\n" % (id1, id2, mat[i][j]))

    def py_matrix(path, ids, mat):
        with open(path, "w") as out:
            out.write("\t%s\n" % "\t".join(ids))
            for i, id1 in enumerate(ids):
                out.write("%s\t%s\n" % (id1, "\t".join(map(str, [int(x) for x in mat[i]]))))

    a, b = str(tmp_path / "a.tsv"), str(tmp_path / "b.tsv")
    for n in (0, 1, 3, 150):
        ids = ["S%d_é" % i if i % 7 == 3 else "sample%d" % i for i in range(n)]
        mat = np.random.default_rng(n).integers(0, 2 ** 31 - 1, size=(n, n), dtype=np.int32) if n else np.zeros((0, 0), dtype=np.int32)
        if n > 1:
            mat[0, 1], mat[1, 0] = 0, -5
        for ours, theirs in ((distance.write_pairwise, py_pairwise), (distance.write_matrix, py_matrix)):
            ours(a, ids, mat)
            theirs(b, ids, mat)
            assert open(a, "rb").read() == open(b, "rb").read(), (n, ours.__name__)
    # a name longer than the writer's buffer, and a strided view of a larger matrix
    ids = ["x" * (5 << 20), "y"]
    big = np.arange(16, dtype=np.int32).reshape(4, 4)
    distance.write_matrix(a, ids, big[:2, :2])
    py_matrix(b, ids, big[:2, :2])
    assert open(a, "rb").read() == open(b, "rb").read()
    with pytest.raises(IOError):
        distance.write_pairwise(str(tmp_path / "no_such_dir" / "p.tsv"), ["a"], np.zeros((1, 1), dtype=np.int32))
    # the bundled files: ids and values from the pairwise fixture reproduce both fixture files byte for byte
    tree, _ = fixture_trees["lambdaVirus"]
    want_p = open(os.path.join(tree, "snp_distance_pairwise.tsv"), "rb").read()
    want_m = open(os.path.join(tree, "snp_distance_matrix.tsv"), "rb").read()
    rows = [ln.split("\t") for ln in want_p.decode().splitlines()[1:]]
    ids = sorted({r[0] for r in rows})
    mat = np.zeros((len(ids), len(ids)), dtype=np.int32)
    for s1, s2, v in rows:
        mat[ids.index(s1), ids.index(s2)] = int(v)
    distance.write_pairwise(a, ids, mat)
    distance.write_matrix(b, ids, mat)
    assert open(a, "rb").read() == want_p and open(b, "rb").read() == want_m


def test_varscan_host_finish_reproduces_every_bundled_var_flt_vcf_line():
    """The product's host half of phase-1 site calling (csrc/varscan_rows.hip through varscan.format_rows): header text and every
    data line of the 58 bundled var.flt.vcf files from the line's own counts; the ExtraParams parser; the strand filter,
    allele choice and --p-value on hand-made records."""
    import tarfile
    import numpy as np
    from oracle import varscan_oracle as vo
    from snp_pipeline_amd import varscan
    from snp_pipeline_amd.device import VARSCAN_DTYPE
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fixtures")
    opts = varscan.Options("--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5")
    n = 0
    for ds in ("lambdaVirus", "agona", "listeria"):
        with tarfile.open(os.path.join(here, ds, "expected.tar.xz")) as t:
            for m in t.getmembers():
                if not m.name.endswith("/var.flt.vcf"):
                    continue
                lines = t.extractfile(m).read().decode().splitlines(True)
                assert "".join(ln for ln in lines if ln.startswith("#")) == varscan.header_text(15)
                data = [ln for ln in lines if not ln.startswith("#")]
                # one record and one stand-in pileup line (chrom, position, then anything) per fixture line
                recs = np.zeros(len(data), dtype=VARSCAN_DTYPE)
                pile = bytearray()
                for k, ln in enumerate(data):
                    f = ln.rstrip("\n").split("\t")
                    v = dict(zip(f[8].split(":"), f[9].split(":")))
                    rd, ad = int(v["RD"]), int(v["AD"])
                    total = rd + ad                                          # FREQ's denominator: indel reads count, DP does not matter
                    while vo.java_percent(ad, total) != v["FREQ"]:
                        total += 1
                        assert total < 4 * (rd + ad) + 64, ln
                    r = recs[k]
                    r["line_off"], r["sdp"], r["dp"], r["total"] = len(pile), int(v["SDP"]), int(v["DP"]), total
                    r["rdf"], r["rdr"], r["adf"], r["adr"] = int(v["RDF"]), int(v["RDR"]), int(v["ADF"]), int(v["ADR"])
                    r["ref_qual_sum"], r["alt_qual_sum"] = int(v["RBQ"]) * rd, int(v["ABQ"]) * ad
                    r["ref_base"], r["alt_base"] = ord(f[3]), ord(f[4])
                    pile += ("%s\t%s\tN\t0\t*\t*\n" % (f[0], f[1])).encode()
                text, n_rows = varscan.format_rows(recs, bytes(pile), opts)
                assert n_rows == len(data) and text.decode() == "".join(data), m.name
                n += n_rows
    assert n == 69019
    o = varscan.Options("--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5")
    assert (o.min_coverage, o.min_reads2, o.min_avg_qual, o.min_var_freq, o.min_freq_for_hom, o.p_value, o.strand_filter) == (8, 5, 15, 0.9, 0.75, 0.99, 1)
    o = varscan.Options("--output-vcf 1 --p-value 1e-3 --strand-filter 0 --min-coverage 3 --variants --min-freq-for-hom 0.8")
    assert (o.min_coverage, o.p_value, o.strand_filter, o.min_freq_for_hom, o.min_reads2) == (3, 1e-3, 0, 0.8, 2)
    p = o.device_params()
    assert (p.min_coverage, p.min_reads2, p.min_avg_qual, p.min_var_freq) == (3, 2, 15, 0.2)

    # hand-made records: (rdf, rdr, adf, adr, total) -> FILTER / GT; two alleles on one line; --p-value
    def one(rdf, rdr, adf, adr, total, alt="G", off=0):
        r = np.zeros(1, dtype=VARSCAN_DTYPE)
        r["line_off"], r["sdp"], r["dp"], r["total"], r["rdf"], r["rdr"], r["adf"], r["adr"] = off, total, total, total, rdf, rdr, adf, adr
        r["ref_qual_sum"], r["alt_qual_sum"], r["ref_base"], r["alt_base"] = 30 * (rdf + rdr), 31 * (adf + adr), ord("A"), ord(alt)
        return r
    line = b"ctg\t77\tA\t40\t...\tIII\n"
    dflt = varscan.Options("")
    row = lambda recs, op=dflt: varscan.format_rows(recs, line, op)[0].decode().split("\t")
    assert row(one(10, 10, 0, 20, 40))[6] == "str10" and row(one(1, 0, 0, 20, 21))[6] == "PASS" and row(one(20, 0, 0, 20, 40))[6] == "PASS"
    assert row(one(10, 10, 10, 10, 40))[6] == "PASS" and row(one(10, 10, 0, 20, 40), varscan.Options("--strand-filter 0"))[6] == "PASS"
    assert row(one(10, 10, 10, 10, 40))[9].startswith("0/1:") and row(one(2, 2, 20, 16, 40))[9].startswith("1/1:")
    assert row(one(2, 2, 20, 16, 40))[:6] == ["ctg", "77", ".", "A", "G", "."]
    both = np.concatenate([one(0, 0, 5, 5, 30, "C"), one(0, 0, 10, 10, 30, "T")])
    assert row(both)[4] == "T" and row(np.concatenate([one(0, 0, 5, 5, 30, "C"), one(0, 0, 5, 5, 30, "T")]))[4] == "C"
    assert varscan.format_rows(one(1000, 1000, 1, 1, 2002), line, varscan.Options("--p-value 0.05")) == (b"", 0)      # p = 0.75
    assert varscan.format_rows(np.zeros(0, dtype=VARSCAN_DTYPE), b"", dflt) == (b"", 0)


def test_call_sites_input_errors_follow_the_reference_protocol(tmp_path, monkeypatch, capsys):
    """call_sites.py:53-60: a missing / empty reference is a global error (exit 100), a missing BAM a sample error (exit 100, or 98
    with StopOnSampleError=false); the BAM name follows RemoveDuplicateReads / EnableLocalRealignment (call_sites.py:55-59).
    Nothing here touches the device."""
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    log = tmp_path / "error.log"
    monkeypatch.setenv("errorOutputFile", str(log))
    sdir = tmp_path / "samples" / "s1"
    sdir.mkdir(parents=True)
    ref = tmp_path / "ref.fasta"

    def run():
        args = cli.parse_command_line("call_sites -v 0 %s %s" % (ref, sdir))
        with pytest.raises(SystemExit) as ei:
            cli.run_command_from_args(args)
        return ei.value.code

    monkeypatch.delenv("StopOnSampleError", raising=False)
    assert run() == 100 and "Reference file" in log.read_text()
    ref.write_text(">c\nACGT\n")
    log.write_text("")
    assert run() == 100 and "reads.sorted.deduped.indelrealigned.bam" in log.read_text()
    monkeypatch.setenv("StopOnSampleError", "false")
    monkeypatch.setenv("RemoveDuplicateReads", "false")
    log.write_text("")
    assert run() == 98 and "reads.sorted.indelrealigned.bam" in log.read_text()
    monkeypatch.setenv("EnableLocalRealignment", "false")
    log.write_text("")
    assert run() == 98 and "Sample BAM file %s/reads.sorted.bam" % sdir in log.read_text()
    capsys.readouterr()


def test_library_fasta_matrix_loader_equals_the_line_loop(tmp_path):
    """csrc/fasta_in.hip (snp_matrix.load_matrix) against read_matrix, the statement of distance.py:76-84: wrapped and unwrapped
    sequences, CR LF and lone CR line ends, '>>' headers, an empty id, empty records, duplicate ids, no final newline, blanks
    kept, unequal lengths padded with '-', text before the first header."""
    import random
    import numpy as np
    from snp_pipeline_amd import snp_matrix
    rng = random.Random(5)
    path = str(tmp_path / "m.fasta")
    cases = [b">a\nACGT\nAC\n>b\nA\n", b">>x y\r\nAC\rGT\r\n>\n\n>x y\nTTTT", b"", b">only\n", b">a\nAC\n>a\nGGG\n>c\n \n", b">\xc3\xa9\nAC", b"\n"]
    for _ in range(40):
        recs = []
        for r in range(rng.randint(1, 12)):
            name = rng.choice(["s%d" % r, "dup", ">odd", "with space", ""])
            seq = "".join(rng.choice("ACGTacgt-N .") for _ in range(rng.choice((0, 1, 59, 60, 61, 200))))
            width = rng.choice((60, 7, 10 ** 6))
            eol = rng.choice(("\n", "\n", "\r\n", "\r"))
            body = eol.join(seq[k:k + width] for k in range(0, len(seq), width))
            recs.append(">" + name + eol + body + (eol if rng.random() < 0.8 else ""))
        text = "".join(recs)
        cases.append(text.encode())
    for data in cases:
        with open(path, "wb") as f:
            f.write(data)
        try:
            want = snp_matrix.read_matrix(path)
        except UnboundLocalError:
            with pytest.raises(UnboundLocalError):
                snp_matrix.load_matrix(path)
            continue
        ids, mat, lens = snp_matrix.load_matrix(path)
        got = {}
        for r, i in enumerate(ids):                             # later records of the same id replace earlier ones, as in the dict
            got[i] = bytes(mat[r, :int(lens[r])]).decode("utf-8")
            assert bytes(mat[r, int(lens[r]):]) == b"-" * (mat.shape[1] - int(lens[r]))
        assert got == want, data
        assert mat.shape == (len(ids), max([len(v.encode()) for v in want.values()] + [0]) if ids else 0) or len(set(ids)) != len(ids)
    with open(path, "wb") as f:
        f.write(b"ACGT\n>a\nAC\n")
    with pytest.raises(UnboundLocalError):
        snp_matrix.load_matrix(path)
    with pytest.raises(IOError):
        snp_matrix.load_matrix(str(tmp_path / "absent.fasta"))


def test_snp_matrix_whole_file_copy_equals_the_text_mode_line_loop(tmp_path):
    """snp_matrix.py:112-117 copies the consensus files line by line in text mode; the whole-file copy gives the same bytes for
    CR LF / lone CR line ends, a missing final newline and non-ASCII text, in sorted directory order, and a missing file is the
    reference's sample error."""
    import argparse
    from snp_pipeline_amd import snp_matrix
    blobs = [b">a\nACGT\n", b">b\r\nAC\rGT\r\n", b">c\nno final newline", "é>x\n".encode("utf-8"), b"\n\n"]
    dirs = []
    for i, b in enumerate(blobs):
        sd = tmp_path / ("s%d" % (9 - i))
        sd.mkdir()
        (sd / "consensus.fasta").write_bytes(b)
        dirs.append(str(sd))
    (tmp_path / "dirs.txt").write_text("\n".join(dirs) + "\n")
    args = argparse.Namespace(sampleDirsFile=str(tmp_path / "dirs.txt"), consFileName="consensus.fasta", snpmaFile=str(tmp_path / "snpma.fasta"),
                              forceFlag=True, verbose=0)
    snp_matrix.create_snp_matrix(args)
    want = []
    for sd in sorted(dirs):
        with open(os.path.join(sd, "consensus.fasta"), "r", encoding="utf-8") as f:
            want.extend(f)
    assert (tmp_path / "snpma.fasta").read_bytes() == "".join(want).encode("utf-8")
    (tmp_path / "s9" / "consensus.fasta").write_bytes(b">a\n\xff\n")
    with pytest.raises(UnicodeDecodeError):
        snp_matrix.create_snp_matrix(args)


def test_library_vcf_site_reader_and_snplist_writer(tmp_path, fixture_trees):
    """csrc/vcf_in.hip: the CHROM / POS arrays of every bundled VCF equal read_vcf_sites' tuples; files outside the plain case fall
    back to the Python reader (same sites, same exceptions); the snplist text equals utils.write_list_of_snps'."""
    import numpy as np
    from snp_pipeline_amd import merge_sites, utils
    n_files = 0
    for ds in ("lambdaVirus", "agona", "listeria"):
        root, _ = fixture_trees[ds]
        for dirpath, _, files in os.walk(root):
            for name in files:
                if name.endswith(".vcf"):
                    path = os.path.join(dirpath, name)
                    want = utils.read_vcf_sites(path)[2]
                    names, cidx, pos = utils.read_vcf_site_arrays(path)
                    assert [(names[int(c)], int(p)) for c, p in zip(cidx, pos)] == want, path
                    n_files += 1
    assert n_files > 70
    odd = {
        "crlf.vcf": b"##fileformat=VCFv4.1\r\n#CHROM\tPOS\r\nc1\t5\tx\r\nc2\t7\r\n\r\nc1\t9",
        "spaces.vcf": b"#CHROM POS\nc1 5 x\nc2\t7\n",                  # a line without TAB: the whitespace split of the Python reader
        "plus.vcf": b"#h\nc1\t+5\tx\nc1\t1_0\n",                        # int() grammar
        "empty_chrom.vcf": b"#h\n\t5\tx\n",
        "header_only.vcf": b"##a\n#CHROM\n",
        "late_header.vcf": b"#h\nc1\t5\n#again\nc9\t6\n",
        "nohdr.vcf": b"c1\t5\nc2\t8\n",                                # no '#' line: the first line is the header (PyVCF), one record
        "meta_only_then_data.vcf": b"##a\nc1\t5\nc2\t8\n",
    }
    for name, data in odd.items():
        path = str(tmp_path / name)
        with open(path, "wb") as f:
            f.write(data)
        want = utils.read_vcf_sites(path)[2]
        names, cidx, pos = utils.read_vcf_site_arrays(path)
        assert [(names[int(c)], int(p)) for c, p in zip(cidx, pos)] == want, name
    assert utils.read_vcf_sites(str(tmp_path / "nohdr.vcf"))[2] == [("c2", 8)]
    for name, data, exc in (("badpos.vcf", b"#h\nc1\tx\n", ValueError),):
        path = str(tmp_path / name)
        with open(path, "wb") as f:
            f.write(data)
        with pytest.raises(exc):
            utils.read_vcf_site_arrays(path)
    # the writer
    contigs = ["ctg|1", "z", "é"]
    uniq = np.array([(0 << 32) | 5, (0 << 32) | 4000000000, (2 << 32) | 1], dtype=np.uint64)
    off = np.array([0, 2, 3, 6], dtype=np.uint32)
    car = np.array([0, 2, 1, 2, 0, 1], dtype=np.uint32)
    samples = ["s0", "sample one", "x"]
    a, b = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
    merge_sites.write_snplist(a, contigs, uniq, off, car, samples)
    utils.write_list_of_snps(b, [(contigs[int(k) >> 32], int(k) & 0xFFFFFFFF) for k in uniq],
                             [[samples[int(i)] for i in car[off[j]:off[j + 1]]] for j in range(len(uniq))])
    assert open(a, "rb").read() == open(b, "rb").read()
    merge_sites.write_snplist(a, [], np.zeros(0, np.uint64), np.zeros(1, np.uint32), np.zeros(0, np.uint32), [])
    assert open(a, "rb").read() == b""


def test_library_snplist_reader_equals_read_snp_position_list(tmp_path, fixture_trees):
    """csrc/vcf_in.hip snpgpu_snplist_sites (utils.read_snp_position_arrays) against utils.read_snp_position_list on the six
    bundled snplists and on lines outside the plain case, which go through the Python reader (same answers, same exceptions)."""
    from snp_pipeline_amd import utils
    n = 0
    for ds in ("lambdaVirus", "agona", "listeria"):
        root, _ = fixture_trees[ds]
        for name in ("snplist.txt", "snplist_preserved.txt"):
            path = os.path.join(root, name)
            want = utils.read_snp_position_list(path)
            names, cidx, pos = utils.read_snp_position_arrays(path)
            assert [(names[int(c)], int(p)) for c, p in zip(cidx, pos)] == want, path
            n += len(want)
    assert n > 14000
    odd = {"spaces.txt": b"c1 5 1 a\nc2\t7\t1\tb\n", "plus.txt": b"c1\t+5\t1\ta\n", "crlf.txt": b"c1\t5\t1\ta\r\nc1\t9\t1\tb\r\n",
           "hash.txt": b"#c\t5\t1\ta\n", "blank_in_name.txt": b"c 1\t5\t1\ta\n", "two_fields.txt": b"c1\t5\nc1\t6"}
    for name, data in odd.items():
        path = str(tmp_path / name)
        with open(path, "wb") as f:
            f.write(data)
        want = utils.read_snp_position_list(path)
        names, cidx, pos = utils.read_snp_position_arrays(path)
        assert [(names[int(c)], int(p)) for c, p in zip(cidx, pos)] == want, name
    empty = str(tmp_path / "empty.txt")
    open(empty, "wb").close()
    assert utils.read_snp_position_arrays(empty)[0] == [] and utils.read_snp_position_list(empty) == []
    for name, data in (("blank.txt", b"c1\t5\t1\ta\n\nc1\t6\t1\ta\n"), ("badpos.txt", b"c1\tx\t1\ta\n"), ("one_field.txt", b"c1\n")):
        path = str(tmp_path / name)
        with open(path, "wb") as f:
            f.write(data)
        with pytest.raises(ValueError):
            utils.read_snp_position_list(path)
        with pytest.raises(ValueError):
            utils.read_snp_position_arrays(path)


def test_library_consensus_file_writer_reproduces_the_lambda_fixtures(tmp_path, fixture_trees):
    """snpgpu_write_consensus_files (csrc/vcf_rows.hip; what hot_path_batch writes its consensus files with): the sixteen
    bundled lambda consensus*.fasta / consensus*.vcf files come back byte for byte from the per-site records their rows
    encode — all samples and both flows in ONE call on host threads; rows in pileup order although the records are handed over
    in site order with shuffled line offsets; the preserved flow's Region rows of positions outside its snplist included."""
    import numpy as np
    from snp_pipeline_amd import _lib as L
    from snp_pipeline_amd import device, utils
    from snp_pipeline_amd.device import COUNTS_DTYPE
    root, _ = fixture_trees["lambdaVirus"]
    samples = sorted(os.listdir(os.path.join(root, "samples")))
    snplists = {sfx: utils.read_snp_position_list(os.path.join(root, "snplist%s.txt" % sfx)) for sfx in ("", "_preserved")}
    # one site set for everything: every position any file mentions
    every = set(p for sl in snplists.values() for _, p in sl)
    parsed = {}
    for sfx in ("", "_preserved"):
        for s in samples:
            rows = [ln for ln in open(os.path.join(root, "samples", s, "consensus%s.vcf" % sfx)).read().split("\n") if ln and not ln.startswith("#")]
            parsed[(s, sfx)] = [_counts_from_vcf_row(r) + (r,) for r in rows]
            every.update(p for _, p, _, _, _, _, _ in parsed[(s, sfx)])
    chrom = snplists[""][0][0]
    keys = np.array(sorted(every), dtype=np.uint64)
    slot = {int(p): i for i, p in enumerate(keys)}

    class Set(object):                      # what write_consensus_files reads of a SiteSet
        _names = np.frombuffer(chrom.encode(), dtype=np.uint8)
        _offs = np.array([0, len(chrom)], dtype=np.uint32)

        def __len__(self):
            return len(keys)
    Set.keys = keys
    names = ["RawDpth", "VarFreq60", "Depth3", "StrDpth0", "StrBias0", "Region"]
    rng = np.random.default_rng(3)
    jobs, want = [], []
    for sfx in ("", "_preserved"):
        in_flow = np.zeros(len(keys), dtype=np.uint8)
        for _, p in snplists[sfx]:
            in_flow[slot[p]] = 1
        for s in samples:
            sdir = os.path.join(root, "samples", s)
            text = open(os.path.join(sdir, "consensus%s.vcf" % sfx)).read()
            header = "".join(ln + "\n" for ln in text.split("\n") if ln.startswith("#"))
            counts = np.zeros(len(keys), dtype=COUNTS_DTYPE)
            line_off = np.zeros(len(keys), dtype=np.uint64)
            row_filters = np.zeros(len(keys), dtype=np.uint8)
            called = {}
            # line offsets: increasing with the row's place in the file, whatever the site order
            offs = np.sort(rng.choice(10 ** 7, size=len(parsed[(s, sfx)]), replace=False)) + 1
            for k, (c_, pos, c, ranked, ft, gt, row) in enumerate(parsed[(s, sfx)]):
                mask = 0
                for name in ([] if ft == "PASS" else ft.split(";")):
                    mask |= 1 << names.index(name)
                i = slot[pos]
                counts[i] = c
                counts[i]["filters"] = mask & ~L.F_REGION           # the record carries the caller's own filters; Region comes with the flow
                row_filters[i] = mask
                line_off[i] = offs[k]
                called[pos] = "-" if (mask or not ranked or ranked[0] == "*") else ranked[0]
            seq = np.frombuffer("".join(called.get(p, "-") for _, p in snplists[sfx]).encode(), dtype=np.uint8)
            out_f, out_v = str(tmp_path / ("%s%s.fasta" % (s, sfx))), str(tmp_path / ("%s%s.vcf" % (s, sfx)))
            jobs.append({"fasta_path": out_f, "fasta_id": s.encode(), "sequence": seq, "vcf_path": out_v, "vcf_header": header.encode(),
                         "counts": counts, "line_off": line_off, "row_filters": row_filters, "site_in_flow": in_flow})
            want.append((out_f, open(os.path.join(sdir, "consensus%s.fasta" % sfx), "rb").read(), out_v, text.encode(), len(parsed[(s, sfx)])))
    res = device.write_consensus_files(jobs, Set(), names, False, ".", n_threads=4)
    assert len(res) == 8
    for (rc, n_rows), (out_f, fasta, out_v, vcf, n) in zip(res, want):
        assert rc == 0 and n_rows == n
        assert open(out_f, "rb").read() == fasta and open(out_v, "rb").read() == vcf
    # an unwritable path is that job's error only; a record with too many symbols is refused, not truncated
    jobs[0]["fasta_path"] = str(tmp_path / "no_such_dir" / "x.fasta")
    jobs[1]["counts"] = jobs[1]["counts"].copy()
    jobs[1]["counts"]["n_symbols"][np.flatnonzero(jobs[1]["counts"]["status"] == L.ST_OK)[0]] = 9
    res = device.write_consensus_files(jobs[:3], Set(), names, False, ".")
    assert [rc for rc, _ in res] == [L.E_IO, L.E_UNSUPPORTED, 0]
    # an empty snplist: ">name" alone, and a VCF of its header (regression_tests.sh:3156-3207 only asks for non-empty files)
    e = {"fasta_path": str(tmp_path / "e.fasta"), "fasta_id": b"s", "sequence": np.zeros(0, np.uint8), "vcf_path": str(tmp_path / "e.vcf"),
         "vcf_header": b"#h\n", "counts": np.zeros(1, dtype=COUNTS_DTYPE), "line_off": np.zeros(1, np.uint64)}

    class Empty(Set):
        def __len__(self):
            return 0
    assert device.write_consensus_files([e], Empty(), names, False, ".") == [(0, 0)]
    assert open(e["fasta_path"], "rb").read() == b">s\n" and open(e["vcf_path"], "rb").read() == b"#h\n"


def test_mpileup2snp_files_reports_an_unwritable_vcf_per_sample_and_never_hangs(monkeypatch):
    """an exception from the VCF writer must not leave the producer thread blocked on its queue (more than ~18 files
    per GPU used to hang the process): the sample carries the error, the others are written, the helper thread is gone."""
    import threading
    from snp_pipeline_amd import varscan

    class FakeDevice(object):
        def varscan_files(self, paths, prm):
            return [([], 10) for _ in paths]
    calls = []

    def writer(vcf_path, pileup_path, records, opts):
        calls.append(vcf_path)
        if vcf_path in ("v3", "v40"):
            raise OSError("No space left on device")
        return 0
    monkeypatch.setattr(varscan, "_write_vcf", writer)
    before = threading.active_count()
    res = varscan.mpileup2snp_files(FakeDevice(), ["p%d" % i for i in range(64)], ["v%d" % i for i in range(64)], varscan.Options(""))
    assert len(calls) == 64 and [i for i, r in enumerate(res) if isinstance(r, Exception)] == [3, 40]
    assert all(r == (10, 0) for i, r in enumerate(res) if i not in (3, 40)) and threading.active_count() == before

    def interrupt(vcf_path, pileup_path, records, opts):
        raise KeyboardInterrupt()
    monkeypatch.setattr(varscan, "_write_vcf", interrupt)
    with pytest.raises(KeyboardInterrupt):
        varscan.mpileup2snp_files(FakeDevice(), ["p%d" % i for i in range(64)], ["v%d" % i for i in range(64)], varscan.Options(""))
    assert threading.active_count() == before


def test_service_runs_the_cli_for_thin_clients(tmp_path, fixture_trees):
    """SNPGPU_SERVICE: the console script hands argv / cwd / environment to the per-node server and prints / exits as the server
    says.  Host-side steps only here (no GPU): outputs, exit codes (0, 2 for argparse, 100 for a global error with its
    error-log entry written where the CLIENT's environment says) equal the in-process run; eight clients at once; a client
    without a reachable server works in-process."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "bin", "cfsan_snp_pipeline")
    sdir = str(tmp_path / "svc")
    env0 = {k: v for k, v in os.environ.items() if k != "SNPGPU_SERVICE"}
    server = subprocess.Popen([sys.executable, exe, "serve", "--socketDir", sdir, "--device", "0", "--idleTimeout", "60"], env=env0,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        deadline = time.time() + 60
        while not os.path.exists(os.path.join(sdir, "dev0.sock")):
            assert server.poll() is None and time.time() < deadline, "the server did not come up"
            time.sleep(0.05)
        lroot, _ = fixture_trees["lambdaVirus"]
        ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fixtures", "lambdaVirus", "lambda_virus.fasta")
        if not os.path.exists(ref):
            ref = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.dirname(lroot)) for f in fs if f.endswith(".fasta") and "lambda" in f.lower() and "snp" not in f.lower()][0]

        def run(argv, service, cwd=str(tmp_path), extra_env=None):
            env = dict(env0)
            if service:
                env["SNPGPU_SERVICE"] = sdir
            env.update(extra_env or {})
            return subprocess.run([sys.executable, exe] + argv, cwd=cwd, env=env, capture_output=True, text=True, timeout=120)

        outs = {}
        for mode in (False, True):
            out = str(tmp_path / ("ref_%d.fasta" % mode))
            r = run(["snp_reference", "-f", "-l", os.path.join(lroot, "snplist.txt"), "-o", out, ref], mode)
            assert r.returncode == 0 and "snp_reference finished" in r.stdout, (r.stdout, r.stderr)
            assert "# Working Directory : %s" % tmp_path in r.stdout                 # the client's cwd, also when the server ran it
            outs[mode] = open(out, "rb").read()
        assert outs[False] == outs[True] == open(os.path.join(lroot, "referenceSNP.fasta"), "rb").read()
        # argparse error: exit 2 and the message on stderr, from the server as in-process
        a, b = run(["snp_reference"], False), run(["snp_reference"], True)
        assert a.returncode == b.returncode == 2 and a.stderr == b.stderr and "Error:" in b.stderr
        # a global error: exit 100, the error log written at the CLIENT's errorOutputFile
        for mode in (False, True):
            log = str(tmp_path / ("error_%d.log" % mode))
            r = run(["snp_reference", "-l", str(tmp_path / "no_such_snplist.txt"), "-o", str(tmp_path / "x.fasta"), ref], mode, extra_env={"errorOutputFile": log})
            assert r.returncode == 100 and "does not exist" in open(log).read(), (r.returncode, r.stderr)
        # eight clients at once
        procs = []
        for k in range(8):
            env = dict(env0, SNPGPU_SERVICE=sdir)
            procs.append(subprocess.Popen([sys.executable, exe, "snp_reference", "-f", "-v", "0", "-l", os.path.join(lroot, "snplist_preserved.txt"),
                                           "-o", str(tmp_path / ("c%d.fasta" % k)), ref], cwd=str(tmp_path), env=env))
        assert [p.wait(timeout=120) for p in procs] == [0] * 8
        want = open(os.path.join(lroot, "referenceSNP_preserved.fasta"), "rb").read()
        assert all(open(str(tmp_path / ("c%d.fasta" % k)), "rb").read() == want for k in range(8))
        # no server behind the directory: in-process
        r = run(["snp_reference", "-f", "-l", os.path.join(lroot, "snplist.txt"), "-o", str(tmp_path / "d.fasta"), ref], False, extra_env={"SNPGPU_SERVICE": str(tmp_path / "nobody")})
        assert r.returncode == 0 and open(str(tmp_path / "d.fasta"), "rb").read() == outs[False]
        r = subprocess.run([sys.executable, exe, "serve", "--socketDir", sdir, "--stop"], env=env0, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and server.wait(timeout=60) == 0
    finally:
        if server.poll() is None:
            server.kill()


def test_exception_hooks_and_metrics_update_follow_the_reference(tmp_path, monkeypatch, capsys):
    """(1) a failed external program (subprocess.CalledProcessError — samtools in call_sites) is logged with its command
    line, not with a Python trace (utils.py:629-700 of the reference); (2) updating an EXISTING metrics file keeps its
    modification time, a new one is created, and concurrent updates do not lose each other."""
    import subprocess
    import threading
    import time
    from snp_pipeline_amd import utils
    log = tmp_path / "error.log"
    monkeypatch.setenv("errorOutputFile", str(log))
    monkeypatch.setenv("StopOnSampleError", "false")
    monkeypatch.setattr("sys.argv", ["cfsan_snp_pipeline", "call_sites", "ref.fasta", "sample1"])
    try:
        raise subprocess.CalledProcessError(1, "samtools mpileup -f ref.fasta reads.bam")
    except subprocess.CalledProcessError:
        import sys
        with pytest.raises(SystemExit) as ei:
            utils.handle_sample_exception(*sys.exc_info())
    assert ei.value.code == 98
    text = log.read_text()
    assert "Error detected while running cfsan_snp_pipeline call_sites." in text
    assert "The error occured while running:\n    samtools mpileup -f ref.fasta reads.bam\n" in text and "exception in function" not in text
    err = capsys.readouterr().err
    assert "Error occured while running:\n    samtools mpileup -f ref.fasta reads.bam" in err and "Traceback" not in err
    try:
        raise KeyError("x")
    except KeyError:
        import sys
        with pytest.raises(SystemExit) as ei:
            utils.handle_global_exception(*sys.exc_info())
    assert ei.value.code == 100 and "KeyError exception in function test_exception_hooks" in log.read_text()
    capsys.readouterr()
    # metrics
    m = tmp_path / "metrics"
    m.write_text("sample=s1\naveInsertSize=250\n")
    old = time.time() - 5000
    os.utime(str(m), (old, old))
    utils.update_properties(str(m), {"missingPos": "7"}, keep_mtime=True)
    assert m.read_text() == "sample=s1\naveInsertSize=250\nmissingPos=7\n" and abs(os.stat(str(m)).st_mtime - old) < 1e-3
    utils.update_properties(str(m), {"missingPos": "8"})
    assert os.stat(str(m)).st_mtime > old + 1000
    fresh = tmp_path / "metrics_new"
    utils.update_properties(str(fresh), {"avePileupDepth": "31.20"}, keep_mtime=True)
    assert fresh.read_text() == "avePileupDepth=31.20\n"
    threads = [threading.Thread(target=utils.update_properties, args=(str(fresh), {"k%d" % i: str(i)}), kwargs={"keep_mtime": True}) for i in range(16)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    got = dict(ln.split("=") for ln in fresh.read_text().splitlines())
    assert got == dict([("avePileupDepth", "31.20")] + [("k%d" % i, str(i)) for i in range(16)])


def test_library_fasta_loader_in_parallel_ranges_equals_the_line_loop(tmp_path):
    """csrc/fasta_in.hip splits a large snpma.fasta into byte ranges, one per thread (2 GB at BASELINE configs[4]): records that
    straddle range bounds, CR LF pairs cut by one, '>' inside sequence text, records without sequence, and a header as the
    very last line must come out as from the line loop of distance.py:76-84 (snp_matrix.read_matrix)."""
    import numpy as np
    from snp_pipeline_amd import snp_matrix
    rng = np.random.default_rng(11)
    letters = np.frombuffer(b"ACGTacgt-N>", dtype=np.uint8)
    for variant, eol in ((0, b"\n"), (1, b"\r\n"), (2, b"\r")):
        parts = []
        n_rec = 330
        for i in range(n_rec):
            L = int(rng.integers(0, 400_000)) if i % 7 else 0
            parts.append(b">" * (1 + i % 3) + b"sample %05d" % i + eol)
            if L:
                row = rng.choice(letters, size=L, p=[.2, .2, .2, .2, .03, .03, .03, .03, .04, .03, .01])
                width = (60, 61, 59)[variant]
                starts = row[::width]
                starts[starts == ord(">")] = ord("A")                 # a '>' at a line start would be a header
                wrapped = b"".join(row[k:k + width].tobytes() + eol for k in range(0, L, width))
                parts.append(wrapped)
        parts.append(b">last_without_sequence")
        data = b"".join(parts)
        assert len(data) > 48 << 20                                   # several ranges
        path = str(tmp_path / ("big%d.fasta" % variant))
        with open(path, "wb") as f:
            f.write(data)
        want = snp_matrix.read_matrix(path)
        ids, mat, lens = snp_matrix.load_matrix(path)
        assert len(ids) == n_rec + 1 and len(set(ids)) == len(ids) and ids[-1] == "last_without_sequence"
        assert mat.shape[1] == max(len(v) for v in want.values())
        for r in (0, 1, 7, 8, 100, 163, 164, 165, 166, 200, 250, 328, 329, 330):
            assert lens[r] == len(want[ids[r]]), (variant, r)
            assert mat[r, :lens[r]].tobytes() == want[ids[r]].encode("latin-1"), (variant, r)
            assert (mat[r, lens[r]:] == 0x2D).all()
        assert [int(x) for x in lens] == [len(want[i]) for i in ids]
        os.remove(path)


def test_library_distance_tsv_writer_in_parallel_row_blocks(tmp_path):
    """csrc/tsv_out.hip formats large matrices in row blocks on several threads (sizes first, then pwrite at the block's place in
    the file): both layouts byte for byte as the print loops of distance.py:100-114 would write them, for ids of unequal
    lengths (one of them empty, one non-ASCII) and values of 1 to 10 digits."""
    import numpy as np
    from snp_pipeline_amd import distance
    rng = np.random.default_rng(4)
    n = 2100                                                   # n * n above the single-thread threshold
    ids = sorted(set(["s%d" % (i * 7919 % 100003) for i in range(n - 3)] + ["", "é_sample", "x" * 300]))
    n = len(ids)
    mat = rng.integers(0, 10 ** rng.integers(1, 10, size=(n, n)), dtype=np.int64).astype(np.int32)
    mat[5, 7] = 2147483647
    np.fill_diagonal(mat, 0)
    a, b = str(tmp_path / "p.tsv"), str(tmp_path / "m.tsv")
    distance.write_pairwise(a, ids, mat)
    distance.write_matrix(b, ids, mat)
    rows = mat.tolist()
    want_p = "Seq1\tSeq2\tDistance\n" + "".join("%s\t%s\t%i\n" % (ids[i], ids[j], rows[i][j]) for i in range(n) for j in range(n))
    want_m = "\t" + "\t".join(ids) + "\n" + "".join("%s\t%s\n" % (ids[i], "\t".join(str(v) for v in rows[i])) for i in range(n))
    assert open(a, "rb").read() == want_p.encode("utf-8")
    assert open(b, "rb").read() == want_m.encode("utf-8")
    # a file that cannot be created is an IOError, also on the threaded path
    with pytest.raises(IOError):
        distance.write_pairwise(str(tmp_path / "no_such_dir" / "p.tsv"), ids, mat)


def test_vcf_split_copies_record_lines_which_is_the_pyvcf_round_trip_for_varscan_output(tmp_path):
    """filter_regions writes var.flt_preserved / _removed.vcf by copying the input's record lines.  The reference sends every
    record through PyVCF3's reader and writer (filter_regions.py:250-275), which re-formats what it parsed as numbers: a
    Float-typed INFO / FORMAT value ("60.00" -> "60.0") and a numeric QUAL.  VarScan's output — the only caller the pipeline
    runs in front of filter_regions — has neither: pinned here on every bundled var.flt.vcf (no Type=Float in any header, QUAL
    '.' in every record, and every bundled var.flt_preserved + _removed pair is a verbatim partition of its var.flt.vcf).
    The second half documents the known difference for other callers' files: the lines are still copied, not re-formatted."""
    import tarfile
    import numpy as np
    from snp_pipeline_amd import filter_regions as fr
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fixtures")
    n_files = n_records = n_pairs = 0
    for ds in ("lambdaVirus", "agona", "listeria"):
        with tarfile.open(os.path.join(here, ds, "expected.tar.xz")) as t:
            members = {m.name: m for m in t.getmembers()}
            for name, m in members.items():
                if not name.endswith("/var.flt.vcf"):
                    continue
                lines = t.extractfile(m).read().decode().splitlines(True)
                header = [ln for ln in lines if ln.startswith("#")]
                records = [ln for ln in lines if not ln.startswith("#")]
                assert not any("Type=Float" in ln for ln in header)
                assert all(ln.split("\t")[5] == "." for ln in records)
                n_files += 1
                n_records += len(records)
                pres, rem = name[:-4] + "_preserved.vcf", name[:-4] + "_removed.vcf"
                if pres in members and rem in members:
                    kept = [ln for ln in t.extractfile(members[pres]).read().decode().splitlines(True) if not ln.startswith("#")]
                    gone = [ln for ln in t.extractfile(members[rem]).read().decode().splitlines(True) if not ln.startswith("#")]
                    assert sorted(kept + gone, key=records.index) == records and not set(kept) & set(gone)
                    n_pairs += 1
    assert n_files >= 50 and n_records > 1000 and n_pairs >= 4
    # another caller's file: MQ is Float-typed; PyVCF3 would write MQ=60.0, the copy keeps the text
    other = tmp_path / "other.vcf"
    other.write_text('##fileformat=VCFv4.1\n##INFO=<ID=MQ,Number=1,Type=Float,Description="mapping quality">\n'
                     "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n"
                     "c1\t5\t.\tA\tG\t30.50\tPASS\tMQ=60.00\tGT\t1\nc1\t9\t.\tC\tT\t.\tPASS\tMQ=7\tGT\t1\n")
    header, data_lines, (names, cidx, pos) = fr._read_vcf(str(other))
    assert list(pos) == [5, 9] and names == ["c1"]
    out = tmp_path / "split.vcf"
    fr._write_vcf(str(out), header, data_lines, np.array([True, False]))
    assert out.read_text().splitlines()[-1] == "c1\t5\t.\tA\tG\t30.50\tPASS\tMQ=60.00\tGT\t1"


def test_a_pileup_that_is_no_regular_file_is_an_input_error_as_a_fifo_is_for_the_reference(tmp_path):
    """pileup.Reader opens the file, closes it, and opens it again to read (pileup.py:401-403, 414): a FIFO does not survive
    that — the writer sees the first reader go away — so the reference cannot consume one either.  Here a pileup path that
    is not a regular file is refused before anything is read (SNPGPU_E_IO per file in the library: csrc/stream.hip), with the
    reference's input-file error protocol at the CLI: nothing blocks on a pipe nobody writes to."""
    import stat
    import subprocess
    import sys
    fifo = tmp_path / "reads.all.pileup"
    os.mkfifo(str(fifo))
    assert stat.S_ISFIFO(os.stat(str(fifo)).st_mode)
    snplist = tmp_path / "snplist.txt"
    snplist.write_text("c1\t5\t1\ts1\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bin", "cfsan_snp_pipeline"), "call_consensus", "-l", str(snplist),
                        "-o", str(tmp_path / "consensus.fasta"), str(fifo)], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, SNPGPU_SERVICE=""))
    assert r.returncode != 0 and not (tmp_path / "consensus.fasta").exists()


def test_subcommands_outside_the_hot_path_go_to_the_reference_cli(tmp_path):
    """With this build's bin/ in front of PATH, `cfsan_snp_pipeline run|map_reads|collect_metrics ...` is handed, argument for
    argument, to the reference's console script: $SNPGPU_REFERENCE_CLI or the next cfsan_snp_pipeline on PATH; without either the
    command ends as a global error (exit 100) that says so."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "bin", "cfsan_snp_pipeline")
    other = tmp_path / "refbin"
    other.mkdir()
    fake = other / "cfsan_snp_pipeline"
    fake.write_text("#!/bin/sh\necho \"reference got: $*\"\nexit 7\n")
    fake.chmod(0o755)
    env = {k: v for k, v in os.environ.items() if not k.startswith("SNPGPU_")}
    env["errorOutputFile"] = str(tmp_path / "error.log")
    on_path = dict(env, PATH=os.pathsep.join([os.path.join(root, "bin"), str(other), env.get("PATH", "")]))
    r = subprocess.run([exe, "map_reads", "-f", "-v", "3", "ref.fasta", "a b.fastq"], env=on_path, capture_output=True, text=True, timeout=120)
    assert (r.returncode, r.stdout) == (7, "reference got: map_reads -f -v 3 ref.fasta a b.fastq\n")
    named = dict(env, SNPGPU_REFERENCE_CLI=str(fake), PATH=os.path.join(root, "bin") + os.pathsep + "/usr/bin" + os.pathsep + "/bin")
    r = subprocess.run([exe, "run", "-s", "samples", "ref.fasta"], env=named, capture_output=True, text=True, timeout=120)
    assert (r.returncode, r.stdout) == (7, "reference got: run -s samples ref.fasta\n")
    alone = dict(env, PATH=os.path.join(root, "bin") + os.pathsep + "/usr/bin" + os.pathsep + "/bin")
    r = subprocess.run([exe, "collect_metrics", "-v", "0", "x"], env=alone, capture_output=True, text=True, timeout=120)
    assert r.returncode == 100 and "not part of the MI355X hot-path build" in r.stderr + r.stdout + open(env["errorOutputFile"]).read()
    # the hot-path subcommands never go there
    r = subprocess.run([exe, "distance", "--version"], env=on_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "reference got" not in r.stdout


def _call_sites_tree(tmp_path):
    """A sample directory whose pileup is fresh (samtools is not run, call_sites.py:70-72) and a reference file."""
    sdir = tmp_path / "samples" / "s1"
    sdir.mkdir(parents=True)
    ref = tmp_path / "ref.fasta"
    ref.write_text(">c\nACGT\n")
    old = time.time() - 1000
    os.utime(str(ref), (old, old))
    bam = sdir / "reads.sorted.deduped.indelrealigned.bam"
    bam.write_bytes(b"placeholder")
    os.utime(str(bam), (old, old))
    (sdir / "reads.all.pileup").write_bytes(b"c\t1\tA\t9\t.........\tIIIIIIIII\n")
    return ref, sdir


def _fake_java(tmp_path, body):
    """A `java` on PATH that records its arguments and prints `body` — the image has no JVM and no VarScan jar; what is checked
    is the command the reference would run (call_sites.py:96-99) and what becomes of its output."""
    bindir = tmp_path / "fakebin"
    bindir.mkdir(exist_ok=True)
    java = bindir / "java"
    java.write_text("#!/bin/sh\necho \"$@\" >> %s/java_calls.txt\ncase \"$*\" in *mpileup2snp*) printf '%%s' '%s';; *) echo 'VarScan v2.3.9' 1>&2;; esac\n" % (tmp_path, body))
    java.chmod(0o755)
    return str(bindir)


def test_call_sites_runs_the_varscan_jar_exactly_as_the_reference_does(tmp_path, monkeypatch, capsys):
    """SNPGPU_SITE_CALLING=varscan, and auto with a jar on CLASSPATH (call_sites.py:89-108): `java <VarscanJvm_ExtraParams> -jar
    <jar> mpileup2snp <pileup> --output-vcf 1 <VarscanMpileup2snp_ExtraParams>` with stdout as var.flt.vcf; no device involved
    (this test runs without one).  Without a jar: the reference's global error in mode varscan."""
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    ref, sdir = _call_sites_tree(tmp_path)
    vcf_text = "##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSample1\nc\t1\t.\tA\tG\t.\tPASS\tADP=9\tGT\t1/1\n"
    monkeypatch.setenv("PATH", _fake_java(tmp_path, vcf_text) + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("CLASSPATH", "/opt/x/picard.jar:/opt/y/VarScan.v2.3.9.jar")
    monkeypatch.setenv("VarscanJvm_ExtraParams", "-Xmx300m")
    monkeypatch.setenv("VarscanMpileup2snp_ExtraParams", "--min-var-freq 0.90")
    log = tmp_path / "error.log"
    monkeypatch.setenv("errorOutputFile", str(log))
    for mode in ("varscan", None):                           # None: auto -> varscan, because the jar is on CLASSPATH
        if mode:
            monkeypatch.setenv("SNPGPU_SITE_CALLING", mode)
        else:
            monkeypatch.delenv("SNPGPU_SITE_CALLING")
        (tmp_path / "java_calls.txt").write_text("")
        if (sdir / "var.flt.vcf").exists():
            (sdir / "var.flt.vcf").unlink()                  # (stale by absence: -f would also re-run samtools)
        cli.run_command_from_args(cli.parse_command_line("call_sites %s %s" % (ref, sdir)))
        out = capsys.readouterr().out
        calls = (tmp_path / "java_calls.txt").read_text().splitlines()
        assert calls[-1] == "-Xmx300m -jar /opt/y/VarScan.v2.3.9.jar mpileup2snp %s/reads.all.pileup --output-vcf 1 --min-var-freq 0.90" % sdir
        assert (sdir / "var.flt.vcf").read_text() == vcf_text
        assert "# Create vcf file" in out and "java -Xmx300m -jar /opt/y/VarScan.v2.3.9.jar mpileup2snp" in out and "# VarScan version v2.3.9" in out
    # fresh: nothing runs
    (tmp_path / "java_calls.txt").write_text("")
    cli.run_command_from_args(cli.parse_command_line("call_sites %s %s" % (ref, sdir)))
    assert (tmp_path / "java_calls.txt").read_text() == "" and "already freshly created" in capsys.readouterr().out
    # the reference's checks of the result (utils.sample_error_on_file_contains)
    monkeypatch.setenv("PATH", _fake_java(tmp_path, "Insufficient memory") + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("StopOnSampleError", "false")
    monkeypatch.setenv("SNPGPU_SITE_CALLING", "varscan")
    (sdir / "var.flt.vcf").unlink()
    with pytest.raises(SystemExit) as ei:
        cli.run_command_from_args(cli.parse_command_line("call_sites %s %s" % (ref, sdir)))
    assert ei.value.code == 98 and "contains unexpected text: 'Insufficient' after running VarScan." in log.read_text()
    # no jar: global error, the reference's words
    monkeypatch.setenv("CLASSPATH", "/opt/x/picard.jar")
    (sdir / "var.flt.vcf").unlink()
    with pytest.raises(SystemExit) as ei:
        cli.run_command_from_args(cli.parse_command_line("call_sites %s %s" % (ref, sdir)))
    assert ei.value.code == 100 and "Error: cannot execute VarScan. Define the path to VarScan.jar in the CLASSPATH environment variable." in log.read_text()
    capsys.readouterr()


def test_site_calling_mode_existing_never_writes_and_batch_modes(tmp_path, monkeypatch, capsys):
    """Mode existing: var.flt.vcf is an input — present and not older than the pileup, else a sample error; call_sites_batch
    --siteCalling varscan runs the jar per stale sample (no device involved)."""
    from snp_pipeline_amd import call_sites as cs
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    ref, sdir = _call_sites_tree(tmp_path)
    log = tmp_path / "error.log"
    monkeypatch.setenv("errorOutputFile", str(log))
    monkeypatch.setenv("StopOnSampleError", "false")
    monkeypatch.setenv("SNPGPU_SITE_CALLING", "existing")
    with pytest.raises(SystemExit) as ei:
        cli.run_command_from_args(cli.parse_command_line("call_sites %s %s" % (ref, sdir)))
    assert ei.value.code == 98 and "needs the var.flt.vcf" in log.read_text()
    (sdir / "var.flt.vcf").write_text("foreign\n")
    stamp = os.stat(str(sdir / "var.flt.vcf")).st_mtime_ns
    cli.run_command_from_args(cli.parse_command_line("call_sites %s %s" % (ref, sdir)))
    assert (sdir / "var.flt.vcf").read_text() == "foreign\n" and os.stat(str(sdir / "var.flt.vcf")).st_mtime_ns == stamp
    late = time.time() + 100
    os.utime(str(sdir / "reads.all.pileup"), (late, late))
    with pytest.raises(SystemExit) as ei:
        cli.run_command_from_args(cli.parse_command_line("call_sites %s %s" % (ref, sdir)))
    assert ei.value.code == 98 and "is older than" in log.read_text()
    # the batch subcommand with the jar
    dirs_file = tmp_path / "dirs.txt"
    dirs_file.write_text("%s\n" % sdir)
    monkeypatch.setenv("PATH", _fake_java(tmp_path, "##fileformat=VCFv4.1\n") + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("CLASSPATH", "/somewhere/varscan.jar")
    monkeypatch.delenv("SNPGPU_SITE_CALLING")
    cli.run_command_from_args(cli.parse_command_line("call_sites_batch --siteCalling varscan %s %s" % (ref, dirs_file)))
    assert (sdir / "var.flt.vcf").read_text() == "##fileformat=VCFv4.1\n"
    assert cs.site_calling_mode() == "varscan" and cs.site_calling_mode("device") == "device"
    monkeypatch.setenv("CLASSPATH", "")
    assert cs.site_calling_mode() == "device"
    monkeypatch.setenv("SNPGPU_SITE_CALLING", "nonsense")
    with pytest.raises(SystemExit) as ei:
        cs.site_calling_mode()
    assert ei.value.code == 100
    capsys.readouterr()


def test_private_directories_and_what_the_service_client_sends(tmp_path, monkeypatch):
    """the per-user directory under a world-writable place is used only when it is provably ours (a real directory,
    owned by this uid, closed to others); the service client sends the variables the steps read, not the whole environment, and
    only to a server of the same uid; the metrics update leaves no lock or temporary file in the sample directory."""
    import socket
    from snp_pipeline_amd import _paths, service, utils
    monkeypatch.delenv("XDG_RUNTIME_DIR", raising=False)
    monkeypatch.setattr("tempfile.tempdir", str(tmp_path))
    d = _paths.private_dir("service")
    assert d == os.path.join(str(tmp_path), "snpgpu-%d" % os.getuid(), "service") and (os.stat(d).st_mode & 0o777) == 0o700
    os.chmod(os.path.dirname(d), 0o777)                       # somebody opened it up: refused
    with pytest.raises(_paths.UnsafeDirectory):
        _paths.private_dir("service")
    os.chmod(os.path.dirname(d), 0o700)
    os.rmdir(d)
    os.symlink(str(tmp_path), d)                              # a link where the directory should be: refused
    with pytest.raises(_paths.UnsafeDirectory):
        _paths.private_dir("service")
    monkeypatch.setenv("SNPGPU_SERVICE", "auto")
    assert service.try_client(["snp_reference", "-h"]) is None          # not used, the caller works in-process
    os.unlink(d)
    xdg = tmp_path / "xdg"
    xdg.mkdir(mode=0o700)
    monkeypatch.setenv("XDG_RUNTIME_DIR", str(xdg))             # (one place with or without a session — a shell and a
    assert _paths.private_dir() == os.path.join(str(tmp_path), "snpgpu-%d" % os.getuid())      # scheduler job must see each other's locks)
    # what travels
    monkeypatch.setenv("AWS_SECRET_ACCESS_KEY", "hunter2")
    monkeypatch.setenv("CallConsensus_ExtraParams", "-q 15")
    monkeypatch.setenv("errorOutputFile", "/x/error.log")
    sent = {k: v for k, v in os.environ.items() if service._forwarded(k)}
    assert "AWS_SECRET_ACCESS_KEY" not in sent and "HOME" not in sent and "SNPGPU_SERVICE" not in sent
    assert sent["CallConsensus_ExtraParams"] == "-q 15" and sent["errorOutputFile"] == "/x/error.log" and "PATH" in sent
    a, b = socket.socketpair(socket.AF_UNIX, socket.SOCK_STREAM)
    try:
        assert service._peer_is_me(a) and service._peer_is_me(b)
    finally:
        a.close()
        b.close()
    # the metrics by-product: locked beside the file (two hosts of a job array both see that lock), replaced in one step,
    # nothing left behind
    sample = tmp_path / "sampleX"
    sample.mkdir()
    utils.update_properties(str(sample / "metrics"), {"missingPos": "3"})
    utils.update_properties(str(sample / "metrics"), {"avePileupDepth": "22.10"}, keep_mtime=True)
    assert sorted(os.listdir(str(sample))) == ["metrics"] and (sample / "metrics").read_text() == "missingPos=3\navePileupDepth=22.10\n"


def test_fasta_header_words_split_as_text_mode_does(tmp_path):
    """str.split() also splits at 0x1c-0x1f; the bytes path has to agree with the text-mode line loop."""
    from snp_pipeline_amd import utils
    p = tmp_path / "odd.fasta"
    p.write_bytes(b">id1\x1c>id2 rest\nAC\x1dGT\n>\x1e\nAA\n")
    fast = utils.fasta_records_ascii(str(p))
    assert fast == [("id1", b"ACGT"), ("", b"AA")]
    slow = {}
    name = None
    with open(str(p), "r") as f:
        for line in f:
            if line.startswith(">"):
                words = line[1:].split()
                name = words[0] if words else ""
                slow[name] = 0
            else:
                slow[name] += len("".join(line.split()))
    assert {k: len(v) for k, v in fast} == slow


def test_contig_names_escape_to_ascii_in_order_and_back(tmp_path, monkeypatch):
    """utf8_names.py: the escape the device sees instead of non-ASCII contig names is injective, keeps the bytewise order, leaves
    every other column alone, works across its read blocks, and refuses what text and bytes would read differently."""
    import random
    from snp_pipeline_amd import utf8_names as u
    rng = random.Random(3)
    names = list({bytes(rng.choice([0x41, 0x7d, 0x7e, 0x7f, 0x80, 0xc3, 0xa4, 0xff, 0x30, 0x3f]) for _ in range(rng.randint(0, 6))) for _ in range(4000)})
    esc = [u.escape_name(n) for n in names]
    assert all(e.isascii() for e in esc) and len(set(esc)) == len(names)
    assert [u.unescape_name(e) for e in sorted(esc)] == sorted(names)
    assert u.escape_names(["chrä", "x"]) == ["chr~45~26", "x"]
    lines = []
    for i in range(3000):
        name = ("chrä", "染色体1", "a~b", "plain")[i % 4]
        lines.append(("%s%s\t%d\tA\t3\t.~.\t~I~" % ("  " if i % 50 == 0 else "", name, i + 1)).encode("utf-8") + (b"\r\n" if i % 7 == 0 else (b"\r" if i % 11 == 0 else b"\n")))
    data = b"".join(lines)
    path = tmp_path / "n.pileup"
    path.write_bytes(data)
    monkeypatch.setattr(u, "CHUNK", 997)                         # many read blocks, ends in the middle of lines and of CR LF pairs
    tmp = u.escaped_copy(str(path), directory=str(tmp_path))
    got = open(tmp, "rb").read()
    os.unlink(tmp)
    assert got.isascii() and got.count(b"\n") == data.count(b"\n") and got.count(b"\r") == data.count(b"\r")
    import re
    want = re.sub(rb"(?m)^( *)([^\t]+)", lambda m: m.group(1) + u.escape_name(m.group(2)), data.replace(b"\r\n", b"\n").replace(b"\r", b"\n"))
    assert got.replace(b"\r\n", b"\n").replace(b"\r", b"\n") == want
    assert b".~.\t~I~" in got                                    # a '~' outside the name column stays as it is
    for bad, exc in ((("c\t1\tä\t1\t.\tI\n").encode(), u.Refused), (("c x\t1\tA\t1\t.\tI\n").encode(), u.Refused),
                     (("c\t1\tA\t1\t.\tä\n").encode(), u.Refused), (b"c\xff\t1\tA\t1\t.\tI\n", UnicodeDecodeError)):
        path.write_bytes(bad)
        with pytest.raises(exc):
            u.escaped_copy(str(path), directory=str(tmp_path))
        assert [n for n in os.listdir(str(tmp_path)) if n.startswith("snpgpu_names_")] == []
    vcf = tmp_path / "c.vcf"
    vcf.write_bytes(b"##x\n#CHROM\tPOS\nchr~45~26\t5\t.\tA~\n~00\t6\t.\tC\n")
    u.unescape_vcf_chrom(str(vcf))
    assert vcf.read_bytes() == "##x\n#CHROM\tPOS\nchrä\t5\t.\tA~\n~\t6\t.\tC\n".encode("utf-8")


def test_library_line_rows_equal_python_rows():
    """snpgpu_format_line_rows (the host formatter behind `call_consensus --vcfAllPos`: CHROM and POS from the pileup line itself, the
    numbers from 32-byte line records, the full record where a line is wide) against the row-by-row Python statement of the same layout
    (vcf_writer.row_from_counts <- vcf_writer.py:295-379) on random records and on pileup text with the first two columns in every
    spelling split() and int() take: leading blanks, several blanks between the fields, '+7', '007', '1_000', '-12', '-0', names of one
    byte and of 900, CR LF ends.  Also device.pack_line_records / expand_line_records there and back."""
    import random
    import numpy as np
    from snp_pipeline_amd import _lib as L
    from snp_pipeline_amd import device as dev
    from snp_pipeline_amd import vcf_writer
    rng = random.Random(11)
    names = ["RawDpth", "VarFreq60", "Depth3", "StrDpth0", "StrBias0", "Region"]
    n = 600
    counts = np.zeros(n, dtype=dev.COUNTS_DTYPE)
    flags = np.zeros(n, dtype=np.uint8)
    lines, offs, heads = [], [], []
    at = 0
    for i in range(n):
        c = counts[i]
        k = rng.choice([0, 0, 1, 1, 2, 2, 3, 3, 4, 6, 8])
        syms = rng.sample(list(b"*ACGTN#<"), k)
        big = rng.random() < 0.05                                       # a count past 16 bits: wide
        tot = sorted((rng.randint(1, 70000 if big else 300) for _ in range(k)), reverse=True)
        c["ref_base"] = rng.choice(list(b"ACGTNacgtn"))
        c["raw_depth"] = rng.randint(0, 4_000_000_000) if rng.random() < 0.1 else rng.randint(0, 400)
        c["n_symbols"] = k
        c["status"] = L.ST_OK
        c["cons_base"] = syms[0] if k else ord("-")
        c["filters"] = rng.randint(0, 63) if rng.random() < 0.5 else 0
        for r in range(k):
            f = rng.randint(0, tot[r])
            # (a symbol between 'Z' and 'a' is on neither strand, pileup.py:269-274: forward + reverse may stay below the total)
            c["sym"][r], c["total"][r], c["fwd"][r], c["rev"][r] = syms[r], tot[r], f, tot[r] - f - (1 if tot[r] - f > 0 and rng.random() < 0.05 else 0)
        c["good_depth"] = int(c["total"].sum())
        c["fwd_good_depth"], c["rev_good_depth"] = int(c["fwd"].sum()), int(c["rev"].sum())
        flags[i] = rng.choice([0, 0, 1, 3])
        name = rng.choice(["c", "ctg_7", "gi|9626243|ref|NC_001416.1|", "NODE_1_length_419034_cov_23.1", "n" * 900])
        value = rng.choice([0, 5, 17, 99_999, 4_000_000_000, 12_345_678_901_234_567_890])
        pos_text, pos_int = rng.choice([("%d" % value, value), ("+%d" % value, value), ("00%d" % value, value), ("-%d" % value, -value),
                                        ("1_000" if value else "0_0", 1000 if value else 0)])
        lead = rng.choice(["", "", " ", "\t "])
        sep = rng.choice(["\t", "\t", "  ", " \t"])
        end = rng.choice(["\n", "\n", "\r\n"])
        text = "%s%s%s%s\tA\t3\t.,.\tIII%s" % (lead, name, sep, pos_text, end)
        lines.append(text)
        offs.append(at + 1)
        heads.append((name, pos_int))
        at += len(text)
    pileup = "".join(lines).encode()
    offs = np.array(offs, dtype=np.uint64)
    recs, widx, wide = dev.pack_line_records(counts, flags)
    assert 0 < len(widx) < n // 2 and (recs["n_symbols"][widx] == dev.LINE_WIDE).all()
    assert set(widx.tolist()) == {i for i in range(n) if int(counts[i]["n_symbols"]) > 3 or int(counts[i]["total"].max()) > 65535}
    flags2, back = dev.expand_line_records(recs, widx, wide)
    assert np.array_equal(flags2, flags) and back.tobytes() == counts.tobytes()
    for gt, preserve, only_listed in ((".", False, False), ("0", True, False), ("1", False, True)):
        got, n_rows = vcf_writer.format_line_rows(pileup, offs, recs, widx, wide, names, preserve, gt, only_listed=only_listed)
        keep = [i for i in range(n) if flags[i] or not only_listed]
        want = "".join(vcf_writer.row_from_counts(heads[i][0], heads[i][1], counts[i], names, preserve, gt) + "\n" for i in keep)
        assert n_rows == len(keep)
        if got.decode() != want:
            g, w = got.decode().split("\n"), want.split("\n")
            k = next(j for j in range(len(w)) if j >= len(g) or g[j] != w[j])
            raise AssertionError("row %d:\n%r\n%r" % (k, g[k] if k < len(g) else None, w[k]))
    # a range of the file on its own (what a piece of the file-to-file writer is), and nothing at all
    part, n_part = vcf_writer.format_line_rows(pileup, offs[100:250], recs[100:250], [w - 100 for w in widx if 100 <= w < 250],
                                               wide[[k for k, w in enumerate(widx) if 100 <= w < 250]], names, False, ".")
    assert n_part == 150 and part.decode() == "".join(vcf_writer.row_from_counts(heads[i][0], heads[i][1], counts[i], names, False, ".") + "\n" for i in range(100, 250))
    assert vcf_writer.format_line_rows(pileup, offs[:0], recs[:0], widx[:0], wide[:0], names, False, ".") == (b"", 0)
    # an offset outside the text, a wide line without its record: refused
    with pytest.raises(ValueError):
        vcf_writer.format_line_rows(pileup, np.array([len(pileup) + 5], dtype=np.uint64), recs[:1], widx[:0], wide[:0], names, False, ".")
    k = int(widx[0])
    with pytest.raises(ValueError):
        vcf_writer.format_line_rows(pileup, offs[k:k + 1], recs[k:k + 1], widx[:0], wide[:0], names, False, ".")

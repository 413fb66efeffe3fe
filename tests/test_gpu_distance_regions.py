"""HIP distance / region / site-merge kernels (through the C ABI) against the oracle, golden vectors and the
reference's bundled ExpectedResults."""
import os
import random

import numpy as np
import pytest

from oracle import steps_oracle as so

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def d():
    from tests.gpu_util import get_device
    return get_device()


def _np_distance(sym):
    up = np.where((sym >= 97) & (sym <= 122), sym - 32, sym)
    valid = np.isin(up, np.frombuffer(b"ACGT", dtype=np.uint8))
    n = len(sym)
    out = np.zeros((n, n), dtype=np.int32)
    for i in range(n):
        out[i] = ((up != up[i]) & valid & valid[i]).sum(axis=1)
    return out


def test_distance_golden_pairs(d, steps_vectors):
    for v in steps_vectors["sequence_distance"]:
        if not v["a"]:
            continue
        sym = np.frombuffer((v["a"] + v["b"]).encode(), dtype=np.uint8).reshape(2, -1)
        out = d.distance(sym)
        assert out[0, 1] == v["d"] and out[1, 0] == v["d"] and out[0, 0] == 0 and out[1, 1] == 0


@pytest.mark.parametrize("ds", ["lambdaVirus", "agona", "listeria"])
def test_distance_bundled_fixtures(d, fixture_trees, ds):
    root, _ = fixture_trees[ds]
    for suffix in ("", "_preserved"):
        seqs = so.parse_snpma(open(os.path.join(root, "snpma%s.fasta" % suffix)).read())
        ids = sorted(seqs)
        sym = np.frombuffer("".join(seqs[i] for i in ids).encode(), dtype=np.uint8).reshape(len(ids), -1)
        out = d.distance(sym)
        dd = {(a, b): int(out[i, j]) for i, a in enumerate(ids) for j, b in enumerate(ids)}
        assert so.matrix_text(ids, dd) == open(os.path.join(root, "snp_distance_matrix%s.tsv" % suffix)).read()


@pytest.mark.parametrize("n,s,seed", [(1, 1, 0), (3, 31, 1), (5, 32, 2), (7, 33, 3), (130, 1000, 4), (257, 4097, 5), (300, 20000, 6)])
def test_distance_random_vs_numpy(d, n, s, seed):
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"ACGTacgt-NnRY*", dtype=np.uint8)
    probs = np.array([.2, .2, .2, .2, .03, .03, .03, .03, .04, .01, .01, .005, .005, .01])
    sym = rng.choice(alphabet, size=(n, s), p=probs / probs.sum()).astype(np.uint8)
    out = d.distance(sym)
    assert np.array_equal(out, _np_distance(sym))


def test_distance_tile_sharding_and_properties(d):
    """Cyclic tile assignment over ranks reproduces the full matrix; symmetry and zero diagonal at a larger size."""
    import torch
    rng = np.random.default_rng(8)
    n, s = 700, 3000
    sym = rng.choice(np.frombuffer(b"ACGT-", dtype=np.uint8), size=(n, s), p=[.24, .24, .24, .24, .04]).astype(np.uint8)
    full = d.distance(sym)
    assert np.array_equal(full, full.T) and not full.diagonal().any()
    d.use_torch_stream()
    t = torch.from_numpy(sym).cuda()
    packed = torch.empty(n * d.packed_row_bytes(s), dtype=torch.uint8, device="cuda")
    d.pack_matrix_dev(t.data_ptr(), n, s, s, packed.data_ptr())
    acc = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    for r in range(3):
        part = torch.zeros((n, n), dtype=torch.int32, device="cuda")
        d.distance_packed_dev(packed.data_ptr(), n, s, part.data_ptr(), r, 3)
        acc += part
    torch.cuda.synchronize()
    assert np.array_equal(acc.cpu().numpy(), full)
    sub = rng.choice(n, size=40, replace=False)
    assert np.array_equal(full[np.ix_(sub, sub)], _np_distance(sym[sub]))


def _dense_gpu(d, m, w, snps):
    if not snps:
        return []
    s, e, g = d.dense_windows(snps, [0, len(snps)], [m], [w])
    og, os_, oe = d.merge_regions(g, s, e)
    return [[int(a), int(b)] for a, b in zip(os_, oe)]


def test_region_golden_vectors(d, steps_vectors):
    for v in steps_vectors["find_dense_regions"]:
        assert _dense_gpu(d, v["m"], v["w"], v["snps"]) == v["out"], v
    for v in steps_vectors["merge_regions"]:
        regs = v["in"]
        og, os_, oe = d.merge_regions([0] * len(regs), [r[0] for r in regs], [r[1] for r in regs])
        assert [[int(a), int(b)] for a, b in zip(os_, oe)] == v["out"]
    for v in steps_vectors["in_region"]:
        regs = v["regions"]
        got = d.in_regions([0], [v["pos"]], [0, len(regs)], [r[0] for r in regs], [r[1] for r in regs])
        assert bool(got[0]) == v["out"]


def test_region_pipeline_vs_golden_collect(d, steps_vectors):
    """collect_dense_regions over several samples + merge (mode all), through the host mirror's region builder."""
    from snp_pipeline_amd import filter_regions as fr
    for v in steps_vectors["collect_all"]:
        samples = [[(c, p) for c, p in recs] for recs in v["samples"]]
        got = fr.compute_bad_regions(d, samples, v["lens"], v["edge"], v["max_snps"], v["windows"])
        assert {c: [list(map(int, r)) for r in regs] for c, regs in got.items()} == v["out"]


def test_merge_sites_vs_oracle(d):
    rng = random.Random(4)
    samples = []
    for i in range(9):
        recs = [(rng.choice(["ctgA", "ctgB", "c"]), rng.randint(1, 300)) for _ in range(rng.randint(0, 120))]
        recs += recs[:5]                                  # duplicate records inside one VCF
        samples.append(("dir%02d" % i, "s%02d" % i, recs))
    merged, _ = so.merge_sites(samples)
    contigs = sorted({c for _, _, recs in samples for c, _ in recs})
    cid = {c: i for i, c in enumerate(contigs)}
    keys, samp = [], []
    for i, (_, _, recs) in enumerate(samples):
        for c, p in recs:
            keys.append((cid[c] << 32) | p)
            samp.append(i)
    uniq, off, car = d.merge_sites(keys, samp)
    got = [((contigs[int(k) >> 32], int(k) & 0xFFFFFFFF), ["s%02d" % j for j in car[off[i]:off[i + 1]]]) for i, k in enumerate(uniq)]
    assert got == merged
    u0, o0, c0 = d.merge_sites([], [])
    assert len(u0) == 0 and list(o0) == [0]


def test_sort_scan_primitives_at_scale(d):
    """The hand-written sort / scan path (csrc/prims.h) on inputs that span many tiles and merge passes: unsorted positions
    in many segments, 10^5 random intervals in many groups (with unbounded ends), 3 x 10^5 site records — against numpy /
    the oracle."""
    rng = np.random.default_rng(11)
    # dense windows: 300 segments of 0..3000 unsorted positions with duplicates, three rules
    seg_sizes = rng.integers(0, 3000, size=300)
    seg_sizes[7] = 0
    seg_off = np.concatenate([[0], np.cumsum(seg_sizes)]).astype(np.uint32)
    pos = rng.integers(1, 200_000, size=int(seg_off[-1])).astype(np.int64)
    rules_m, rules_w = [3, 2, 1], [1000, 125, 15]
    cs, ce, cg = d.dense_windows(pos, seg_off, rules_m, rules_w)
    want = []
    for sgi in range(300):
        p_sorted = np.sort(pos[seg_off[sgi]:seg_off[sgi + 1]])
        for i in range(len(p_sorted)):
            for m, w in zip(rules_m, rules_w):
                if i + m < len(p_sorted) and p_sorted[i] + w - 1 >= p_sorted[i + m]:
                    want.append((int(p_sorted[i]), int(p_sorted[i + m]), sgi))
    assert list(zip(cs.tolist(), ce.tolist(), cg.tolist())) == want
    # already sorted input takes the copy path of the sort: same answer
    pos_sorted = np.concatenate([np.sort(pos[seg_off[i]:seg_off[i + 1]]) for i in range(300)]).astype(np.int64)
    cs2, ce2, cg2 = d.dense_windows(pos_sorted, seg_off, rules_m, rules_w)
    assert cs2.tolist() == cs.tolist() and ce2.tolist() == ce.tolist() and cg2.tolist() == cg.tolist()
    # merge_regions: 10^5 intervals, 40 groups, some reaching to the unknown-contig-length sentinel
    n = 100_000
    grp = rng.integers(0, 40, size=n).astype(np.uint32)
    st = rng.integers(0, 3_000_000, size=n).astype(np.int64)
    en = st + rng.integers(0, 60, size=n)
    en[rng.integers(0, n, size=20)] = np.iinfo(np.int64).max
    mg, ms, me = d.merge_regions(grp, st, en)
    want = []
    for g in range(40):
        sel = grp == g
        for a, b in so.merge_regions(sorted(zip(st[sel].tolist(), en[sel].tolist()))):
            want.append((g, a, b))
    assert list(zip(mg.tolist(), ms.tolist(), me.tolist())) == want
    # in_regions against the merged list
    reg_off = np.searchsorted(mg, np.arange(41)).astype(np.uint32)
    q_g = rng.integers(0, 42, size=50_000).astype(np.uint32)
    q_p = rng.integers(0, 3_000_100, size=50_000).astype(np.int64)
    got = d.in_regions(q_g, q_p, reg_off, ms, me)
    for k in rng.integers(0, 50_000, size=3000):
        g, p = int(q_g[k]), int(q_p[k])
        inside = g < 40 and any(a <= p <= b for a, b in zip(ms[reg_off[g]:reg_off[g + 1]].tolist(), me[reg_off[g]:reg_off[g + 1]].tolist()) if a <= p)
        assert bool(got[k]) == inside
    # merge_sites: 3 x 10^5 records of 700 samples over 2 contigs, shuffled, with duplicates
    m = 300_000
    keys = ((rng.integers(0, 2, size=m).astype(np.uint64) << np.uint64(32)) | rng.integers(1, 60_000, size=m).astype(np.uint64))
    samp = rng.integers(0, 700, size=m).astype(np.uint32)
    uniq, off, car = d.merge_sites(keys, samp)
    pairs = np.unique(np.stack([keys, samp.astype(np.uint64)], axis=1), axis=0)
    wu, wc = np.unique(pairs[:, 0], return_counts=True)
    assert uniq.tolist() == wu.tolist()
    assert off.tolist() == np.concatenate([[0], np.cumsum(wc)]).tolist()
    assert car.tolist() == pairs[:, 1].tolist()


def test_small_steps_device_pointer_forms(d):
    """The _dev entry points (device pointers, asynchronous, counts in device memory) give what the host forms give."""
    import torch
    d.use_torch_stream()
    rng = np.random.default_rng(12)
    m = 50_000
    keys = ((rng.integers(0, 3, size=m).astype(np.int64) << 32) | rng.integers(1, 20_000, size=m).astype(np.int64))
    samp = rng.integers(0, 90, size=m).astype(np.int32)
    uniq, off, car = d.merge_sites(keys.astype(np.uint64), samp.astype(np.uint32))
    tk, ts = torch.from_numpy(keys).cuda(), torch.from_numpy(samp).cuda()
    ou = torch.zeros(m, dtype=torch.int64, device="cuda")
    oo = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
    oc = torch.zeros(m, dtype=torch.int32, device="cuda")
    on = torch.zeros(4, dtype=torch.int32, device="cuda")
    d.merge_sites_dev(tk.data_ptr(), ts.data_ptr(), m, ou.data_ptr(), oo.data_ptr(), oc.data_ptr(), on.data_ptr())
    torch.cuda.synchronize()
    nu, nc = int(on[0]), int(on[1])
    assert (nu, nc) == (len(uniq), len(car))
    assert ou[:nu].cpu().numpy().astype(np.uint64).tolist() == uniq.tolist()
    assert oo[:nu + 1].cpu().numpy().tolist() == off.tolist() and oc[:nc].cpu().numpy().tolist() == car.tolist()
    # dense windows + merge + in_regions chained on the device, no host round trip in between
    seg_off = np.array([0, 4000, 4000, 9000, 15000], dtype=np.int32)
    pos = rng.integers(1, 400_000, size=15000).astype(np.int64)
    cs, ce, cg = d.dense_windows(pos, seg_off.astype(np.uint32), [3, 1], [1000, 15])
    tp, tso = torch.from_numpy(pos).cuda(), torch.from_numpy(seg_off).cuda()
    cap = 15000 * 2
    ws_, we_ = torch.zeros(cap, dtype=torch.int64, device="cuda"), torch.zeros(cap, dtype=torch.int64, device="cuda")
    wg_, wn = torch.zeros(cap, dtype=torch.int32, device="cuda"), torch.zeros(2, dtype=torch.int32, device="cuda")
    d.dense_windows_dev(tp.data_ptr(), tso.data_ptr(), 4, 15000, [3, 1], [1000, 15], ws_.data_ptr(), we_.data_ptr(), wg_.data_ptr(), wn.data_ptr())
    torch.cuda.synchronize()
    k = int(wn[0])
    assert int(wn[1]) == 0 and k == len(cs)
    assert ws_[:k].cpu().tolist() == cs.tolist() and we_[:k].cpu().tolist() == ce.tolist() and wg_[:k].cpu().tolist() == cg.tolist()
    mg, ms, me = d.merge_regions(cg, cs, ce)
    og = torch.zeros(k, dtype=torch.int32, device="cuda")
    os_, oe = torch.zeros(k, dtype=torch.int64, device="cuda"), torch.zeros(k, dtype=torch.int64, device="cuda")
    mn = torch.zeros(2, dtype=torch.int32, device="cuda")
    d.merge_regions_dev(wg_.data_ptr(), ws_.data_ptr(), we_.data_ptr(), k, og.data_ptr(), os_.data_ptr(), oe.data_ptr(), mn.data_ptr())
    torch.cuda.synchronize()
    r = int(mn[0])
    assert r == len(mg) and og[:r].cpu().tolist() == mg.tolist() and os_[:r].cpu().tolist() == ms.tolist() and oe[:r].cpu().tolist() == me.tolist()
    # error bits instead of exceptions
    bad = torch.tensor([5, -1, 7], dtype=torch.int64, device="cuda")
    so2 = torch.tensor([0, 3], dtype=torch.int32, device="cuda")
    d.dense_windows_dev(bad.data_ptr(), so2.data_ptr(), 1, 3, [1], [15], ws_.data_ptr(), we_.data_ptr(), wg_.data_ptr(), wn.data_ptr())
    torch.cuda.synchronize()
    assert int(wn[1]) == 1
    with pytest.raises(Exception):
        d.dense_windows(np.array([5, -1, 7], dtype=np.int64), np.array([0, 3], dtype=np.uint32), [1], [15])


def test_distance_unequal_lengths_and_no_sites(d):
    """utils.calculate_sequence_distance walks range(len(seq1)) (utils.py:1156-1158): a longer second sequence is cut, a
    shorter one raises IndexError.  And a matrix without sites: all distances 0, nothing read."""
    import torch
    from snp_pipeline_amd import distance as dm
    seqs = {"a": "ACGT", "b": "ACCTAA", "c": "TCGTAAGG", "b2": "aCGTAC"}
    ids, mat = dm.distance_matrix(d, seqs)
    assert ids == ["a", "b", "b2", "c"]
    for i, x in enumerate(ids):
        for j, y in enumerate(ids):
            if i < j:
                assert mat[i, j] == mat[j, i] == so.sequence_distance(seqs[x], seqs[y]), (x, y)
    assert so.sequence_distance("ACGT", "TCGTAAGG") == 1
    with pytest.raises(IndexError):
        so.sequence_distance("ACGTA", "ACG")
    with pytest.raises(IndexError):
        dm.distance_matrix(d, {"a": "ACGTA", "b": "ACG"})
    ids, mat = dm.distance_matrix(d, {"x": "", "y": ""})
    assert mat.tolist() == [[0, 0], [0, 0]]
    d.use_torch_stream()
    out = torch.full((300, 300), 7, dtype=torch.int32, device="cuda")
    d.distance_packed_dev(0, 300, 0, out.data_ptr())
    torch.cuda.synchronize()
    assert not bool(out.any().item())


def test_distance_cli_on_an_untidy_snpma(d, tmp_path):
    """The distance subcommand through the native FASTA loader: ids out of order, a duplicate id (the later record counts, as in the
    reference's dict), CR LF line ends, wrapped and unwrapped records, unequal lengths in non-decreasing id order — TSVs equal to
    the oracle's; a shorter later sequence raises IndexError as utils.py:1158 does; text before the first header
    UnboundLocalError (distance.py:84)."""
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    rng = np.random.default_rng(12)
    letters = np.frombuffer(b"ACGTacgt-N", dtype=np.uint8)

    def seq(n):
        return bytes(rng.choice(letters, size=n)).decode()

    recs = [("zeta", seq(500)), ("alpha", seq(300)), ("mid", seq(400)), ("alpha", seq(310)), ("beta", seq(310)), ("omega", seq(500))]
    text = ""
    for k, (name, s_) in enumerate(recs):
        eol = "\r\n" if k % 2 else "\n"
        width = 60 if k % 3 else 10 ** 6
        text += ">" + name + eol + eol.join(s_[i:i + width] for i in range(0, len(s_), width)) + eol
    snpma = tmp_path / "snpma.fasta"
    snpma.write_bytes(text.encode())
    args = cli.parse_command_line("distance -v 0 -p %s/p.tsv -m %s/m.tsv %s" % (tmp_path, tmp_path, snpma))
    assert cli.run_command_from_args(args) == 0
    seqs = so.parse_snpma(text.replace("\r\n", "\n"))
    assert seqs["alpha"] == recs[3][1]
    ids, table = so.distance_tables(seqs)
    assert (tmp_path / "p.tsv").read_text() == so.pairwise_text(ids, table) and (tmp_path / "m.tsv").read_text() == so.matrix_text(ids, table)
    snpma.write_bytes(b">a\nACGTA\n>b\nACG\n")
    with pytest.raises(IndexError):
        cli.run_command_from_args(cli.parse_command_line("distance -f -v 0 -p %s/p.tsv %s" % (tmp_path, snpma)))
    snpma.write_bytes(b"ACGT\n>a\nACGTA\n")
    with pytest.raises(UnboundLocalError):
        cli.run_command_from_args(cli.parse_command_line("distance -f -v 0 -p %s/p.tsv %s" % (tmp_path, snpma)))


def test_distance_cli_reproduces_the_reference_drivers_runs(d, tmp_path):
    """tests/golden/distance_runs.json.gz holds what the reference's own distance driver wrote (or raised) for the untidy SNP
    matrix files of oracle/fuzz.untidy_snpmas (gen_golden.py --only distance): the subcommand writes the same bytes / raises the
    same class."""
    import gzip
    import json
    from oracle import fuzz
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    runs = json.loads(gzip.open(os.path.join(os.path.dirname(__file__), "golden", "distance_runs.json.gz")).read())["runs"]
    texts = dict(fuzz.untidy_snpmas())
    assert len(runs) == len(texts) >= 9
    for run in runs:
        snpma = tmp_path / (run["name"] + ".fasta")
        snpma.write_bytes(texts[run["name"]].encode())
        p, m = tmp_path / (run["name"] + ".p.tsv"), tmp_path / (run["name"] + ".m.tsv")
        args = cli.parse_command_line("distance -f -v 0 -p %s -m %s %s" % (p, m, snpma))
        if "exception" in run:
            with pytest.raises(Exception) as ei:
                cli.run_command_from_args(args)
            assert type(ei.value).__name__ == run["exception"], run["name"]
        else:
            assert cli.run_command_from_args(args) == 0
            assert p.read_text() == run["pairwise"], run["name"]
            assert m.read_text() == run["matrix"], run["name"]

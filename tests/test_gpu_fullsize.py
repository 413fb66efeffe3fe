"""BASELINE.json configs[3] and configs[4] at their stated sizes on one MI355X.

The oracle cannot run at these sizes, so the checks are the size-independent ones: symmetry, zero diagonal, numpy on random
pairs and on the last (partial) tile row / column, tile split over ranks == the full run (distance); line counts = newline
counts, matched lines = sites with a line, a 300-sample batch (two scan groups: 256 + 44) == a 125-sample batch (the per-GPU
shard of configs[3], one group) == per-sample calls, and the oracle on the very lines the device picked (consensus).
"""
import numpy as np
import pytest

from oracle import pileup_oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def d():
    from tests.gpu_util import get_device
    return get_device()


def test_distance_10000_x_200000(d):
    """configs[4]: 10 000 samples x 200 000 sites — a 79 x 79 tile grid whose last tile row / column is partial (16 rows)."""
    import torch
    d.use_torch_stream()
    n, s = 10_000, 200_000
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    lut = torch.tensor(list(b"ACGTacgt-N"), dtype=torch.uint8, device="cuda")
    probs = torch.tensor([.2, .2, .2, .2, .03, .03, .03, .03, .05, .03], device="cuda")
    sym = torch.empty((n, s), dtype=torch.uint8, device="cuda")
    chunk = (1 << 28) // s
    for r0 in range(0, n, chunk):
        r1 = min(n, r0 + chunk)
        sym[r0:r1] = lut[torch.multinomial(probs, (r1 - r0) * s, replacement=True, generator=g)].view(r1 - r0, s)
    # two rows made equal, two made maximally different, so that 0 and large values are both present off the diagonal
    sym[17] = sym[4242]
    pk = torch.empty((n, d.packed_row_bytes(s)), dtype=torch.uint8, device="cuda")
    d.pack_matrix_dev(sym.data_ptr(), n, s, s, pk.data_ptr())
    dm = torch.full((n, n), -1, dtype=torch.int32, device="cuda")
    d.distance_packed_dev(pk.data_ptr(), n, s, dm.data_ptr())
    torch.cuda.synchronize()
    assert int(dm.min().item()) == 0 and int(dm.max().item()) < s
    assert not bool(dm.diagonal().any().item())
    assert bool(torch.equal(dm, dm.t()))
    assert int(dm[17, 4242].item()) == 0
    # numpy on 200 random pairs, on the partial last tile row / column, and on tile-boundary rows
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def np_dist(a, b):
        ua, ub = a & 0xDF, b & 0xDF                          # upper-case letters; '-' (0x2D) stays outside ACGT either way
        return int((np.isin(ua, acgt) & np.isin(ub, acgt) & (ua != ub)).sum())

    pairs = [(int(i), int(j)) for i, j in rng.integers(0, n, size=(200, 2))]
    pairs += [(n - 1, 0), (n - 1, n - 2), (9984, 9983), (9984, n - 1), (9999, 9984), (127, 128), (128, 255), (0, 9990)]
    rows = sorted({i for pr in pairs for i in pr})
    host = {i: sym[i].cpu().numpy() for i in rows}
    for i, j in pairs:
        assert int(dm[i, j].item()) == np_dist(host[i], host[j]), (i, j)
    # the tiles split over three ranks, written into one zeroed matrix, are the full matrix
    dm3 = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    for r in range(3):
        d.distance_packed_dev(pk.data_ptr(), n, s, dm3.data_ptr(), r, 3)
    torch.cuda.synchronize()
    assert bool(torch.equal(dm3, dm))


def test_configs3_shard_125_and_grouped_batch_300(d):
    """configs[3] sample shape (5 Mbp x 30x, 50 k sites): 300 samples resident (130 GB) in one call = two scan groups."""
    import torch
    from snp_pipeline_amd import device as dev
    from snp_pipeline_amd import _lib as L
    G, S, B = 5_000_000, 50_000, 300
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
    for i in range(1, B):
        offs[i] = offs[i - 1] + (sizes[i - 1] + 255) // 256 * 256 + (3 if i % 7 == 0 else 0)      # some odd start addresses
    pile = torch.empty(int(offs[-1]) + sizes[-1] + 64, dtype=torch.uint8, device="cuda")
    for i in range(B):
        assert d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), pile.data_ptr() + int(offs[i]), sizes[i]) == sizes[i]
    keys = [(b"synth_chr1", int(p_)) for p_ in pos]
    ss = d.siteset(keys, [L.SITE_IN_SNPLIST] * S)
    p = po.CallerParams(0, 0.6, 3, 0, 0.0)
    prm = dev.make_params(p.min_base_quality, p.min_cons_freq, p.min_cons_depth, p.min_cons_strand_depth, p.min_cons_strand_bias)
    sizes_np = np.asarray(sizes, dtype=np.uint64)
    bases = torch.zeros((B, S), dtype=torch.uint8, device="cuda")
    filt = torch.zeros((B, S), dtype=torch.uint8, device="cuda")
    status = torch.zeros((B, 4), dtype=torch.int64, device="cuda")
    d.call_consensus_batch_dev(ss, pile.data_ptr(), offs, prm, bases.data_ptr(), filt.data_ptr(), status.data_ptr(), sizes=sizes_np)
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    assert (st[:, 0] == -1).all()
    assert not bool((filt & 0x80).any().item())
    for i in range(B):
        view = pile[int(offs[i]):int(offs[i]) + sizes[i]]
        assert int(st[i, 1]) == int((view == 10).sum().item()), i          # lines seen = terminators in the file
    called = (bases != 0x2D).sum(dim=1).cpu().numpy()
    assert (st[:, 2] >= called).all() and (called > 0.9 * S).all()        # every called site had a line
    # the per-GPU shard of configs[3] (125 samples, one scan group) gives the same rows
    b125 = torch.zeros((125, S), dtype=torch.uint8, device="cuda")
    f125 = torch.zeros((125, S), dtype=torch.uint8, device="cuda")
    s125 = torch.zeros((125, 4), dtype=torch.int64, device="cuda")
    d.call_consensus_batch_dev(ss, pile.data_ptr(), offs[:125], prm, b125.data_ptr(), f125.data_ptr(), s125.data_ptr(), sizes=sizes_np[:125])
    torch.cuda.synchronize()
    assert torch.equal(b125, bases[:125]) and torch.equal(f125, filt[:125]) and s125.cpu().numpy().tolist() == st[:125].tolist()
    # two of the samples as FILES through the streamed ingestion (432 MB each: 26 chunks of 16 MiB, two device slots)
    import os
    import tempfile
    with tempfile.TemporaryDirectory(prefix="snpfull_") as tmp:
        paths = []
        for i in (3, 299):
            path = os.path.join(tmp, "s%d.pileup" % i)
            with open(path, "wb") as fh:
                fh.write(pile[int(offs[i]):int(offs[i]) + sizes[i]].cpu().numpy().tobytes())
            paths.append(path)
        results, rcs, stats = d.call_consensus_files(ss, paths, prm, want_line_offsets=True)
        assert list(rcs) == [0, 0] and stats.bytes == sizes[3] + sizes[299] and stats.n_chunks >= 50
        for r, i in zip(results, (3, 299)):
            assert bytes(r.bases) == bytes(bases[i].cpu().numpy()) and bytes(r.filters) == bytes(filt[i].cpu().numpy())
            assert r.status.astype(np.int64).tolist() == st[i].tolist()
            assert int(np.count_nonzero(r.line_offsets)) == int(st[i, 2])
    # single-sample calls around the group boundary and at the ends; the oracle on the lines the device picked
    for i in (0, 124, 255, 256, 257, 299):
        b1 = torch.zeros(S, dtype=torch.uint8, device="cuda")
        f1 = torch.zeros(S, dtype=torch.uint8, device="cuda")
        s1 = torch.zeros(4, dtype=torch.int64, device="cuda")
        d.call_consensus_dev(ss, pile.data_ptr() + int(offs[i]), sizes[i], prm, b1.data_ptr(), f1.data_ptr(), s1.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(b1, bases[i]) and torch.equal(f1, filt[i]) and s1.cpu().numpy().tolist() == st[i].tolist()
        line_off = d.line_offsets(ss)
        assert int(st[i, 2]) == int(np.count_nonzero(line_off))
        bh, fh = bases[i].cpu().numpy(), filt[i].cpu().numpy()
        for slot in rng.choice(S, size=120, replace=False):
            if line_off[slot] == 0:
                assert bh[slot] == 0x2D and fh[slot] == 0
                continue
            a0 = int(offs[i]) + int(line_off[slot]) - 1
            raw = bytes(pile[a0:a0 + 700].cpu().numpy())
            fields = po.split_fields(raw[:raw.index(b"\n")])
            assert (fields[0], int(fields[1])) == keys[slot]
            base, mask = po.call_record(po.parse_record(fields, p.min_base_quality), p)
            want = 0x2D if (mask or base == 0x2A) else base
            assert (int(bh[slot]), int(fh[slot])) == (want, mask)


def test_merge_sites_at_configs4_scale(d):
    """The C1 / merge_sites union at configs[4] size: 10 000 samples x 1 500 records over 200 000 sites on several contigs (15 M
    keys, duplicates inside a sample included) against numpy: unique keys, carrier offsets, carriers in sample order."""
    n, per, S = 10_000, 1_500, 200_000
    rng = np.random.default_rng(8)
    pool = np.unique(rng.integers(1, 5_000_000, size=S, dtype=np.uint64) | (rng.integers(0, 7, size=S, dtype=np.uint64) << np.uint64(32)))
    keys = pool[rng.integers(0, len(pool), size=n * per)]
    samp = np.repeat(np.arange(n, dtype=np.uint32), per)
    perm = rng.permutation(n * per)                              # records arrive in any order
    uniq, off, car = d.merge_sites(keys[perm], samp[perm])
    pairs = np.unique(np.stack([keys, samp.astype(np.uint64)], axis=1), axis=0)       # (key, sample) once each, sorted by key then sample
    want_u, first = np.unique(pairs[:, 0], return_index=True)
    assert np.array_equal(uniq, want_u)
    assert np.array_equal(off, np.append(first, len(pairs)).astype(np.uint32))
    assert np.array_equal(car, pairs[:, 1].astype(np.uint32))


def test_region_steps_at_configs4_scale(d):
    """K3 at configs[4] size: 10 000 segments (samples) x 1 500 positions, the pipeline's three dense-window rules, the interval
    union per group and the classification of all 15 M positions, against numpy restatements of filter_regions.py:17-71 and
    utils.py:1168-1318 (window i is dense iff p[i+M] exists and p[i] + W - 1 >= p[i+M]; sort, join overlapping or adjacent
    intervals; inclusive ends)."""
    n, per = 10_000, 1_500
    rng = np.random.default_rng(10)
    # clustered positions so that windows are dense here and there: cluster centres + small offsets
    centres = rng.integers(1000, 4_999_000, size=(n, per // 5), dtype=np.int64)
    pos = (np.repeat(centres, 5, axis=1) + rng.integers(0, 2500, size=(n, per), dtype=np.int64))
    shuffled = rng.permuted(pos, axis=1)                         # the device sorts each segment itself
    seg_off = (np.arange(n + 1, dtype=np.uint64) * per).astype(np.uint32)
    max_snps, windows = [3, 2, 1], [1000, 125, 15]
    cs, ce, cseg = d.dense_windows(shuffled.reshape(-1), seg_off, max_snps, windows)
    srt = np.sort(pos, axis=1)
    want = []
    for m, w in zip(max_snps, windows):
        a, b = srt[:, :-m], srt[:, m:]
        hit = a + w - 1 >= b
        seg = np.broadcast_to(np.arange(n, dtype=np.int64)[:, None], hit.shape)[hit]
        want.append(np.stack([seg, a[hit], b[hit]], axis=1))
    want = np.concatenate(want)
    got = np.stack([cseg.astype(np.int64), cs, ce], axis=1)
    assert len(got) == len(want) > 1_000_000

    def rows_sorted(x):
        return x[np.lexsort((x[:, 2], x[:, 1], x[:, 0]))]
    assert np.array_equal(rows_sorted(got), rows_sorted(want))
    # union per group (group = segment % 500: many samples feed one group, as mode all does per contig)
    grp = (cseg % 500).astype(np.uint32)
    mg, ms, me = d.merge_regions(grp, cs, ce)
    order = np.lexsort((ce, cs, grp))
    g, s, e = grp[order].astype(np.int64), cs[order], ce[order]
    key_e = e + g * (1 << 40)                                    # running maximum of the end, restarted per group
    run = np.maximum.accumulate(key_e) - g * (1 << 40)
    new = np.ones(len(g), dtype=bool)
    new[1:] = (g[1:] != g[:-1]) | (s[1:] > run[:-1] + 1)
    starts = np.flatnonzero(new)
    want_g, want_s = g[starts], s[starts]
    want_e = run[np.append(starts[1:], len(g)) - 1]
    assert np.array_equal(mg.astype(np.int64), want_g) and np.array_equal(ms, want_s) and np.array_equal(me, want_e)
    # classification of every position against the regions of its group
    reg_off = np.zeros(501, dtype=np.uint32)
    np.cumsum(np.bincount(want_g, minlength=500), out=reg_off[1:])
    pos_group = np.repeat(np.arange(n, dtype=np.uint32) % 500, per)
    flags = d.in_regions(pos_group, shuffled.reshape(-1), reg_off, ms, me)
    p = shuffled.reshape(-1)
    key_s = want_s + want_g * (1 << 40)
    idx = np.searchsorted(key_s, p + pos_group.astype(np.int64) * (1 << 40), side="right") - 1
    inside = (idx >= 0) & (want_g[np.maximum(idx, 0)] == pos_group) & (p <= want_e[np.maximum(idx, 0)])
    assert np.array_equal(flags, inside) and 0.01 < inside.mean() < 0.99


def test_all_positions_vcf_of_a_full_size_sample(d, tmp_path):
    """`call_consensus --vcfAllPos` at BASELINE's sample shape (5 Mbp x 30x: 5 M lines, 76 pieces of the read-back, 470 MB of rows),
    through the file-to-file writer.  Size-independent properties: a row per line, in file order (POS strictly ascending, CHROM the
    one contig), the SDP column adds up to the depth-column sum of the scan (collect_metrics.py:325-340), RD + sum(AD) of every row is
    the good depth the 32-byte records carry; the records equal the 128-byte ones; a sample of rows spread over the file equals the
    row-by-row Python writer."""
    import argparse
    import torch
    from snp_pipeline_amd import device as dev
    from snp_pipeline_amd import _lib as L
    from snp_pipeline_amd import vcf_writer
    G, S = 5_000_000, 50_000
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
    n = d.synth_pileup_dev(3, 0, G, ref.data_ptr(), alt.data_ptr(), 0, 0)
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    assert d.synth_pileup_dev(3, 0, G, ref.data_ptr(), alt.data_ptr(), buf.data_ptr(), n) == n
    text = buf[:n].cpu().numpy().tobytes()
    del buf
    path, out = str(tmp_path / "reads.all.pileup"), str(tmp_path / "consensus.vcf")
    with open(path, "wb") as f:
        f.write(text)
    args = argparse.Namespace(minBaseQual=0, minConsFreq=0.6, minConsDpth=3, minConsStrdDpth=0, minConsStrdBias=0.0, vcfRefName="ref.fasta",
                              vcfPreserveRefCase=False, vcfFailedSnpGt=".")
    prm = dev.make_params(0, 0.6, 3, 0, 0.0)
    ss = d.siteset([(b"synth_chr1", int(p_)) for p_ in pos], [L.SITE_IN_SNPLIST] * S)
    n_lines, n_rows = vcf_writer.write_all_positions_vcf_from_pileup(d, ss, out, "s0", args, path, prm)
    assert n_lines == n_rows == text.count(b"\n") > 4_900_000
    off, recs, widx, wide = d.call_all_lines_compact(ss, path, prm, capacity=n_lines, wide_capacity=n_lines // 100)
    off_full, flags_full, counts_full = d.call_all_lines(ss, path, prm, capacity=n_lines, check=True)
    flags, counts = dev.expand_line_records(recs, widx, wide)
    assert np.array_equal(off, off_full) and np.array_equal(flags, flags_full) and counts.tobytes() == counts_full.tobytes()     # (no spill records in this file)
    assert int(flags.astype(bool).sum()) >= 0.97 * S and len(widx) < n_lines // 1000
    del counts_full, off_full, flags_full
    # the file: the scan's depth-column sum, one row per line in order
    res = d.call_consensus_files(ss, [path], prm, want_depth_sum=True)[0][0]
    sdp_sum, good_sum, prev, rows = 0, 0, 0, 0
    with open(out, "rb") as f:
        for ln in f:
            if ln[0] == 35:                                             # '#'
                continue
            c = ln.split(b"\t")
            p_ = int(c[1])
            assert c[0] == b"synth_chr1" and p_ > prev
            prev = p_
            v = c[9].split(b":")
            sdp_sum += int(v[1])
            good_sum += int(v[2]) + (0 if v[3] == b"0" and c[4] == b"." else sum(int(x) for x in v[3].split(b",")))
            rows += 1
    assert rows == n_lines and sdp_sum == res.depth_sum == int(counts["raw_depth"].astype(np.int64).sum())
    assert good_sum == int(counts["good_depth"].astype(np.int64).sum())
    # rows spread over the file against the Python writer
    names = [nm for nm, _ in vcf_writer.filter_descriptions(0.6, 3, 0, 0.0)]
    pick = np.unique(np.concatenate([np.arange(0, n_lines, 997), widx[:200].astype(np.int64), np.arange(n_lines - 50, n_lines)]))
    want = {}
    for k in pick:
        o = int(off[k]) - 1
        f0, f1 = text[o:o + 64].split(None, 2)[:2]
        want[int(k)] = vcf_writer.row_from_counts(f0.decode(), int(f1), counts[k], names, False, ".").encode() + b"\n"
    with open(out, "rb") as f:
        k = 0
        for ln in f:
            if ln[0] == 35:
                continue
            if k in want:
                assert ln == want[k], k
            k += 1

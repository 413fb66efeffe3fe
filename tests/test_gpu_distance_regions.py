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

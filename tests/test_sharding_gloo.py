"""The N > 1 path on CPU: two gloo ranks shard samples, exchange site lists and matrix rows, split the distance
tiles, and must reproduce the single-process answer.  (The kernels themselves are covered by the -m gpu tests; here
the per-rank arithmetic is the oracle's.)"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import steps_oracle as so
from snp_pipeline_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _np_tile(sym, bi, bj):
    t = sharding.DIST_TILE
    a, b = sym[bi * t:(bi + 1) * t], sym[bj * t:(bj + 1) * t]
    valid = np.isin(sym, np.frombuffer(b"ACGT", dtype=np.uint8))
    va, vb = valid[bi * t:(bi + 1) * t], valid[bj * t:(bj + 1) * t]
    return ((a[:, None, :] != b[None, :, :]) & va[:, None, :] & vb[None, :, :]).sum(axis=2).astype(np.int32)


def _worker(rank, world, port, n_samples, n_sites, seed, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(seed)
        # every rank derives the same global inputs, then only touches its own shard
        site_sets = [np.unique(rng.integers(1, 5000, size=rng.integers(0, 60))).astype(np.int64) for _ in range(n_samples)]
        sym = rng.choice(np.frombuffer(b"ACGT-n", dtype=np.uint8), size=(n_samples, n_sites)).astype(np.uint8)
        lo, hi = sharding.shard_bounds(n_samples, rank, world)
        # C1: variable-length all-gather of site keys, then the same merge everywhere
        mine = np.concatenate([site_sets[i] for i in range(lo, hi)] + [np.zeros(0, np.int64)])
        owner = np.concatenate([np.full(len(site_sets[i]), i, np.int64) for i in range(lo, hi)] + [np.zeros(0, np.int64)])
        keys, counts = sharding.all_gather_varlen(torch.from_numpy(mine))
        owners, _ = sharding.all_gather_varlen(torch.from_numpy(owner))
        assert sum(counts) == sum(len(s) for s in site_sets)
        merged = {}
        for k, o in zip(keys.tolist(), owners.tolist()):
            merged.setdefault(k, []).append(o)
        want, _ = so.merge_sites([("d%03d" % i, i, [("c", int(p)) for p in site_sets[i]]) for i in range(n_samples)])
        assert [(("c", k), merged[k]) for k in sorted(merged)] == want
        # C2: all-gather of this rank's rows
        full = sharding.all_gather_rows(torch.from_numpy(sym[lo:hi].copy()), n_samples)
        assert np.array_equal(full.numpy(), sym)
        # the same gather straight into a matrix padded to whole tiles
        bands = sharding.RowBands(n_samples, world)
        padded = torch.zeros((max(bands.n_padded, world * ((n_samples + world - 1) // world)), n_sites), dtype=torch.uint8)
        sharding.all_gather_rows_into(torch.from_numpy(sym[lo:hi].copy()), n_samples, padded)
        assert np.array_equal(padded[:n_samples].numpy(), sym) and not padded[n_samples:].any()
        # distance tiles dealt cyclically; the row-band exchange leaves every rank with the complete rows of its band
        part = np.zeros((bands.n_padded, bands.n_padded), dtype=np.int32)
        t = sharding.DIST_TILE
        for bi, bj in sharding.tiles_of_rank(n_samples, rank, world):
            blk = _np_tile(full.numpy(), bi, bj)
            part[bi * t:bi * t + blk.shape[0], bj * t:bj * t + blk.shape[1]] = blk
            if bi != bj:
                part[bj * t:bj * t + blk.shape[1], bi * t:bi * t + blk.shape[0]] = blk.T
        band = bands.exchange(torch.from_numpy(part), rank).numpy()
        r0, r1 = bands.band_rows(rank)
        np.save(os.path.join(out_dir, "band%d.npy" % rank), band[:r1 - r0, :n_samples])
        np.save(os.path.join(out_dir, "rows%d.npy" % rank), np.array([r0, r1]))
        # (the all-reduce combine of round 1 stays available and must agree)
        total = sharding.sum_partial_distances(torch.from_numpy(part.copy())).numpy()[:n_samples, :n_samples]
        assert np.array_equal(total[r0:r1], band[:r1 - r0, :n_samples])
        if rank == 0:
            np.save(os.path.join(out_dir, "sym.npy"), sym)
    finally:
        dist.destroy_process_group()


def test_two_rank_pipeline_matches_single_process(tmp_path):
    n_samples, n_sites = 261, 300                   # 3 x 3 tile grid, uneven shards (131 + 130)
    mp.spawn(_worker, args=(2, _free_port(), n_samples, n_sites, 5, str(tmp_path)), nprocs=2, join=True)
    sym = np.load(str(tmp_path / "sym.npy"))
    total = np.zeros((n_samples, n_samples), dtype=np.int32)
    covered = 0
    for r in range(2):
        r0, r1 = np.load(str(tmp_path / ("rows%d.npy" % r)))
        total[r0:r1] = np.load(str(tmp_path / ("band%d.npy" % r)))
        covered += r1 - r0
    assert covered == n_samples                         # the bands partition the rows
    seqs = [bytes(r).decode() for r in sym]
    for i, j in [(0, 1), (5, 200), (130, 131), (260, 0), (128, 255), (17, 17)]:
        assert total[i, j] == (0 if i == j else so.sequence_distance(seqs[i], seqs[j]))
    assert np.array_equal(total, total.T) and not total.diagonal().any()


def test_shard_bounds_and_tiles():
    assert [sharding.shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [sharding.shard_bounds(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    tiles = sharding.upper_tiles(300)
    assert tiles == [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    got = sorted(t for r in range(4) for t in sharding.tiles_of_rank(300, r, 4))
    assert got == sorted(tiles)
    assert sharding.all_gather_varlen(torch.arange(3))[1] == [3]
    # the exchange plan: every block of the full matrix travels exactly once, to the owner of its tile row
    for n, w in ((261, 2), (1000, 8), (129, 3), (5, 4)):
        rb = sharding.RowBands(n, w)
        seen = {}
        for src in range(w):
            for dst in range(w):
                for blk in rb.blocks[src][dst]:
                    assert blk not in seen and rb.bands[dst][0] <= blk[0] < rb.bands[dst][1]
                    seen[blk] = (src, dst)
        assert len(seen) == rb.nt * rb.nt
    # one rank: the exchange is the identity on the padded matrix
    m = torch.arange(256 * 256, dtype=torch.int32).view(256, 256)
    assert torch.equal(sharding.RowBands(200, 1).exchange(m, 0), m)


def test_plans_at_world_8_with_sizes_that_do_not_divide():
    """The exchange plan is a pure function of (n, world): checked here for the 8 ranks of a node without starting any — sample
    blocks cover [0, n) in order; every tile of the upper triangle is computed by exactly one rank; every tile of the FULL matrix
    (a tile and its mirror image) is sent exactly once, to the owner of its tile row; what rank q expects from rank p is what p
    sends to q; the bands cover all rows once."""
    for world in (8, 5, 2):
        for n in (0, 1, 7, 127, 128, 129, 1000, 1023, 1025, 10_000):
            bounds = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == n and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
            assert max(hi - lo for lo, hi in bounds) == (n + world - 1) // world
            dealt = [sharding.tiles_of_rank(n, r, world) for r in range(world)]
            every = sharding.upper_tiles(n)
            assert sorted(t for part in dealt for t in part) == sorted(every) and len(set(every)) == len(every)
            assert max(len(p) for p in dealt) - min(len(p) for p in dealt) <= 1
            bands = sharding.RowBands(n, world)
            nt = bands.nt
            assert bands.n_padded == nt * sharding.DIST_TILE >= n and bands.n_padded - n < sharding.DIST_TILE
            rows = [bands.band_rows(q) for q in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == n and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
            seen = {}
            for src in range(world):
                for dst in range(world):
                    lo, hi = bands.bands[dst]
                    for (r, c) in bands.blocks[src][dst]:
                        assert lo <= r < hi                                  # goes to the owner of its tile row
                        assert ((min(r, c), max(r, c)) in set(dealt[src]))   # and comes from the rank that computed it
                        assert (r, c) not in seen
                        seen[(r, c)] = (src, dst)
            assert len(seen) == nt * nt                                     # the whole matrix, every tile once


def _group_of_one(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", SNPGPU_DIST_AT_WORLD_1="1")
    calls = []
    for name in ("all_gather_into_tensor", "all_to_all_single", "all_reduce"):
        def spy(*a, _f=getattr(dist, name), _n=name, **k):
            calls.append(_n)
            return _f(*a, **k)
        setattr(dist, name, spy)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert sharding.group_of_one_exchanges()
        t = torch.arange(1000, dtype=torch.int64)
        out, counts = sharding.all_gather_varlen(t)
        assert torch.equal(out, t) and counts == [1000]
        out, counts = sharding.all_gather_varlen(t[:0])
        assert out.numel() == 0 and counts == [0]
        rows = torch.randint(0, 255, (300, 77), dtype=torch.uint8)
        assert torch.equal(sharding.all_gather_rows(rows, 300), rows)
        padded = torch.zeros((384, 77), dtype=torch.uint8)
        sharding.all_gather_rows_into(rows, 300, padded)
        assert torch.equal(padded[:300], rows) and not padded[300:].any()
        partial = torch.randint(0, 1000, (384, 384), dtype=torch.int32)
        assert torch.equal(sharding.RowBands(300, 1).exchange(partial, 0), partial)
        assert torch.equal(sharding.sum_partial_distances(partial.clone()), partial)
        assert calls.count("all_gather_into_tensor") == 5 and calls.count("all_to_all_single") == 1 and calls.count("all_reduce") == 1
    finally:
        dist.destroy_process_group()


def test_a_group_of_one_makes_the_calls_when_asked():
    """SNPGPU_DIST_AT_WORLD_1=1: the hook the GPU tests use to run the RCCL entry points on a box with one GPU.  Here over gloo:
    a group of one rank makes every collective call (a group of one without the variable makes none) and gets its own data back."""
    mp.spawn(_group_of_one, args=(_free_port(),), nprocs=1, join=True)
    assert not sharding.group_of_one_exchanges()

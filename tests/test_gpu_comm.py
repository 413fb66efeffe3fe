"""The exchange entry points of the C ABI (csrc/comm.hip) on the one GPU of the test box: a communicator of ONE rank still runs
ncclCommInitRank / ncclAllGather and the library's own plumbing (the self-block copies, the grouped point-to-point calls with no
peer), the tile kernels are checked against numpy, k_group_check against the ATen expressions it replaced.  What needs a second
GPU — ncclSend / ncclRecv between ranks, uneven splits in flight — is covered by construction over gloo (test_gpu_multirank.py,
test_sharding_gloo.py) and measured by the driver's 8-GPU run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def d():
    from tests.gpu_util import get_device
    dev = get_device()
    dev.use_torch_stream()
    return dev


def test_rccl_loads_and_a_communicator_of_one_rank_gathers(d):
    import torch
    assert d.comm_available()
    ver = d.comm_version()
    assert ver and ver >= 20000                                   # NCCL_VERSION_CODE of RCCL 2.x
    uid = d.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    d.comm_init(0, 1, uid)
    try:
        info = d.comm_info()
        assert info == {"rank": 0, "nranks": 1, "rccl_comm_count": 1, "rccl_version": ver}
        src = torch.arange(1000, dtype=torch.int64, device="cuda")
        dst = torch.zeros_like(src)
        d.allgather_dev(src.data_ptr(), dst.data_ptr(), src.numel() * 8)                  # ncclAllGather on the context's stream
        d.stream_wait(10000)
        assert torch.equal(src, dst)
        # blocks of a given size to a given place; the rank's own block is copied by the library
        out = torch.zeros(5000, dtype=torch.uint8, device="cuda")
        blk = torch.arange(777, dtype=torch.int64, device="cuda").to(torch.uint8)
        d.allgatherv_dev(blk.data_ptr(), out.data_ptr(), [777], [1234])
        d.alltoallv_dev(blk.data_ptr(), [777], out.data_ptr() + 3000, [777])
        d.stream_wait(10000)
        o = out.cpu().numpy()
        want = (np.arange(777) % 256).astype(np.uint8)
        assert np.array_equal(o[1234:1234 + 777], want) and np.array_equal(o[3000:3777], want)
        assert not o[:1234].any() and not o[1234 + 777:3000].any() and not o[3777:].any()
        # nothing to send: still a well-formed call
        d.allgatherv_dev(0, out.data_ptr(), [0], [0])
        d.alltoallv_dev(0, [0], 0, [0])
        d.stream_wait(10000)
    finally:
        d.comm_destroy()
    assert d.comm_info()["nranks"] == 1 and d.comm_info()["rccl_comm_count"] == 0
    with pytest.raises(Exception):
        d.allgather_dev(src.data_ptr(), dst.data_ptr(), 8)        # no communicator: refused, not a crash


def test_sharding_routes_through_the_library_when_asked(d, monkeypatch):
    """sharding.use_abi_comm: the three exchanges of the step as library calls, in a group of one (SNPGPU_DIST_AT_WORLD_1), equal
    to what goes in — and to the plain one-rank shortcuts."""
    import torch
    from snp_pipeline_amd import sharding
    monkeypatch.setenv("SNPGPU_DIST_AT_WORLD_1", "1")
    assert sharding.use_abi_comm(d, rank=0, world=1, unique_id=d.comm_unique_id())
    try:
        assert sharding.abi_comm() is d and not sharding._alone()
        keys = torch.arange(12345, dtype=torch.int64, device="cuda") * 7
        got, counts = sharding.all_gather_varlen(keys)
        assert counts == [12345] and torch.equal(got, keys) and got.data_ptr() != keys.data_ptr()
        empty, counts = sharding.all_gather_varlen(keys[:0])
        assert counts == [0] and empty.numel() == 0
        rows = torch.randint(0, 255, (37, 192), dtype=torch.uint8, device="cuda")
        out = torch.zeros((128, 192), dtype=torch.uint8, device="cuda")
        sharding.all_gather_rows_into(rows, 37, out)
        torch.cuda.synchronize()
        assert torch.equal(out[:37], rows) and not out[37:].any()
        assert torch.equal(sharding.all_gather_rows(rows, 37), rows)
        # the row-band exchange: gather kernel -> alltoallv -> scatter kernel gives the matrix back
        bands = sharding.RowBands(300, 1)
        m = torch.randint(-5, 1 << 20, (bands.n_padded, bands.n_padded), dtype=torch.int32, device="cuda")
        band = bands.exchange(m, 0)
        torch.cuda.synchronize()
        assert band.data_ptr() != m.data_ptr() and torch.equal(band, m)
        band2 = bands.exchange(m, 0)                              # the buffers of the plan are reused
        assert band2.data_ptr() == band.data_ptr() and torch.equal(band2, m)
    finally:
        sharding.drop_abi_comm()
    monkeypatch.delenv("SNPGPU_DIST_AT_WORLD_1")
    assert sharding._alone() and sharding.RowBands(300, 1).exchange(m, 0).data_ptr() == m.data_ptr()


def test_tile_kernels_against_numpy(d):
    import torch
    rng = np.random.default_rng(3)
    nt = 5
    n = nt * 128
    m = rng.integers(-1000, 1 << 30, size=(n, n), dtype=np.int32)
    tiles = [(int(a), int(b)) for a, b in rng.integers(0, nt, size=(11, 2))]
    dm = torch.from_numpy(m).cuda()
    rows = torch.tensor([t[0] for t in tiles], dtype=torch.int32, device="cuda")
    cols = torch.tensor([t[1] for t in tiles], dtype=torch.int32, device="cuda")
    packed = torch.zeros((len(tiles), 128, 128), dtype=torch.int32, device="cuda")
    d.tiles_gather_dev(dm.data_ptr(), n, rows.data_ptr(), cols.data_ptr(), len(tiles), packed.data_ptr())
    got = packed.cpu().numpy()
    for k, (a, b) in enumerate(tiles):
        assert np.array_equal(got[k], m[a * 128:(a + 1) * 128, b * 128:(b + 1) * 128]), k
    # ... and back into another matrix, at other places (distinct ones: two tiles to one place would race)
    places = [(k // nt, k % nt) for k in rng.permutation(nt * nt)[:len(tiles)]]
    out = torch.full((n, n), -7, dtype=torch.int32, device="cuda")
    r2 = torch.tensor([p[0] for p in places], dtype=torch.int32, device="cuda")
    c2 = torch.tensor([p[1] for p in places], dtype=torch.int32, device="cuda")
    d.tiles_scatter_dev(packed.data_ptr(), r2.data_ptr(), c2.data_ptr(), len(tiles), out.data_ptr(), n)
    o = out.cpu().numpy()
    want = np.full((n, n), -7, dtype=np.int32)
    for k, (a, b) in enumerate(places):
        want[a * 128:(a + 1) * 128, b * 128:(b + 1) * 128] = got[k]
    assert np.array_equal(o, want)
    d.tiles_gather_dev(dm.data_ptr(), n, rows.data_ptr(), cols.data_ptr(), 0, packed.data_ptr())      # no tiles: nothing happens


@pytest.mark.parametrize("with_counts", [False, True])
def test_group_check_equals_the_expressions_it_replaced(d, with_counts):
    import torch
    from snp_pipeline_amd import _lib as L
    rng = np.random.default_rng(11 + with_counts)
    g, S = 9, 1237
    filt = rng.integers(0, 64, size=(g, S), dtype=np.uint8)
    filt[rng.random((g, S)) < 0.002] |= 0x80
    line = (rng.random((g, S)) < 0.9) * rng.integers(1, 1 << 40, size=(g, S))
    counts = np.zeros((g, S, 128), dtype=np.uint8)
    counts[:, :, 23] = np.where(rng.random((g, S)) < 0.003, L.ST_OK + 1 + rng.integers(0, 3, size=(g, S)), L.ST_OK)
    spilled = rng.random((g, S)) < 0.01
    counts[:, :, 17 + rng.integers(0, 3)][spilled] = 1
    wanted = (rng.random(S) < 0.6).astype(np.uint8)
    excl = [np.sort(rng.choice(S, size=rng.integers(0, 40), replace=False)).astype(np.int32) for _ in range(g)]
    eoff = np.zeros(g + 1, dtype=np.int32)
    np.cumsum([len(e) for e in excl], out=eoff[1:])
    d_filt, d_line, d_counts = torch.from_numpy(filt).cuda(), torch.from_numpy(line.astype(np.int64)).cuda(), torch.from_numpy(counts).cuda()
    d_wanted, d_eoff = torch.from_numpy(wanted).cuda(), torch.from_numpy(eoff).cuda()
    d_es = torch.from_numpy(np.concatenate(excl) if eoff[-1] else np.zeros(1, np.int32)).cuda()
    out = torch.full((g, 3), -1, dtype=torch.int64, device="cuda")
    for own in (True, False):
        d.group_check_dev(0 if with_counts else d_filt.data_ptr(), d_counts.data_ptr() if with_counts else 0, d_line.data_ptr(), d_wanted.data_ptr(),
                          d_eoff.data_ptr() if own else 0, d_es.data_ptr() if own else 0, g, S, out.data_ptr())
        got = out.cpu().numpy()
        bad = (counts[:, :, 23] > L.ST_OK) if with_counts else (filt & 0x80) != 0
        for s in range(g):
            w = wanted.astype(bool).copy()
            if own:
                w[excl[s]] = True
            assert got[s, 0] == int((bad[s] & w).any()), (s, own)
            assert got[s, 1] == int((line[s] != 0).sum())
            assert got[s, 2] == (int((counts[s, :, 17:20] != 0).any(axis=1).sum()) if with_counts else 0)

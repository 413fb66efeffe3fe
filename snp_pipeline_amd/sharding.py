"""Sample sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests).

The path shards by samples: each rank parses its own VCFs and scans its own pileups with no
communication.  There are exactly two exchange steps, both all-gathers:
  C1  the per-rank SNP site keys (variable length) -> every rank runs the same merge -> identical global snplist
  C2  the per-rank rows of the 4-bit packed consensus matrix -> every rank holds the full matrix
The N x N distance is then dealt out in 128 x 128 tiles of the upper triangle, tile t to rank t % world (balanced: every
rank gets the same number of tiles whatever their row).  The combine is a ROW-BAND exchange: rank q ends up with the
complete rows of its contiguous band of tile rows, and every computed tile travels exactly once to the owner of its row
(and its transpose once to the owner of its column) in one all-to-all — N*N*4/world bytes per rank over seven parallel xGMI
links, where summing the partial matrices with an all-reduce would move the whole N x N through every rank.
"""
import os
import sys

import torch
import torch.distributed as dist

DIST_TILE = 128          # csrc/distance.hip

# The device whose context holds this rank's RCCL communicator behind the C ABI (csrc/comm.hip), once use_abi_comm() has set it
# up.  With it every exchange of the step is a library call on the context's stream — ncclAllGather / grouped ncclSend + ncclRecv
# and the hand-written tile kernels — and no ATen kernel runs between the step's own kernels; without it (the gloo route of the
# CPU tests, a host without librccl.so) the exchanges go through torch.distributed as before.
_abi = {"dev": None, "rank": 0, "world": 1, "counts": None}


def use_abi_comm(device, rank=None, world=None, unique_id=None, selftest=True):
    """Make the exchanges of this process library calls on `device`'s context.  With torch.distributed initialised, rank 0's id
    travels through its object broadcast; a host program without PyTorch passes rank / world / the 128-byte id itself (rank 0:
    device.comm_unique_id(); INTEGRATION.md shows the ctypes form).  Returns False when RCCL cannot be loaded, or when the
    communicator's self-test fails on some rank (see _selftest): the exchanges then stay with torch.distributed."""
    via_dist = unique_id is None and dist.is_initialized() and dist.get_world_size() > 1
    available = device.comm_available()
    if via_dist:
        # ncclCommInitRank is itself a collective: a rank that cannot even load librccl.so must say so BEFORE the others walk into it
        available = _all_agree(available)
    if not available:
        return False
    if unique_id is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
        box = [device.comm_unique_id() if rank == 0 else None]
        if dist.is_initialized() and world > 1:
            dist.broadcast_object_list(box, src=0)
        unique_id = box[0]
    if world > 1 and dist.is_initialized():
        # ... and under a watchdog: where one rank fails inside the call the others would wait in it for good, and never reach the
        # agreement below that sends every rank back to torch.distributed
        why = _comm_init_watched(device, rank, world, unique_id, float(os.environ.get("SNPGPU_COMM_INIT_TIMEOUT", "180")))
    else:
        why = None
        device.comm_init(rank, world, unique_id)
    if world > 1 and (selftest or why is not None):
        if why is None:
            why = _selftest(device, int(rank), int(world))
        every_ok = why is None
        if dist.is_initialized():
            flag = torch.tensor([1 if every_ok else 0], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            every_ok = bool(int(flag.item()))
        if not every_ok:
            sys.stderr.write("snpgpu rank %d: the exchanges stay with torch.distributed (%s)\n"
                             % (rank, why or "another rank's self-test of the library's communicator failed"))
            if why is None or not why.startswith("snpgpu_comm_init"):
                device.comm_abort()                              # (a communicator that never came to be has nothing to abort)
            return False
    _abi.update(dev=device, rank=int(rank), world=int(world), counts=None)
    return True


def _all_agree(flag):
    """True where `flag` is true on every rank of the process group (an all-reduce with MIN on the group's own device kind)."""
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def _comm_init_watched(device, rank, world, unique_id, timeout_s):
    """snpgpu_comm_init in a helper thread that this one waits for at most timeout_s.  Returns None, or what went wrong (the call
    raised, or did not come back: its thread is then left behind as a daemon and the communicator is never used)."""
    import threading
    box = {}

    def run():
        try:
            device.comm_init(rank, world, unique_id)
            box["ok"] = True
        except Exception as exc:                                  # noqa: B902 — reported to the other ranks by the caller
            box["err"] = exc

    th = threading.Thread(target=run, name="snpgpu-comm-init", daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        return "snpgpu_comm_init did not return within %g s" % timeout_s
    if "err" in box:
        return "snpgpu_comm_init: %s" % box["err"]
    return None


def _selftest(device, rank, world, timeout_ms=120000):
    """Before the first real exchange, on a stream of its own: one all-gather, one all-gather of blocks of different sizes and one
    all-to-all with a different size for every pair, checked byte for byte.  The first time ranks of a job meet on hardware this
    takes the place of a hang in the middle of a step: a failure (or no completion within the time limit) is reported, the
    communicator is given up and the job goes on over torch.distributed.  Returns None, or what went wrong."""
    side = torch.cuda.Stream()
    try:
        with torch.cuda.stream(side):
            device.use_torch_stream()
            mark = lambda a, b, n: ((torch.arange(n, dtype=torch.int64, device="cuda") * 31 + a * 7919 + b * 104729) & 0xFF).to(torch.uint8)  # noqa: E731
            # 1: equal blocks
            src = mark(rank, 0, 4096)
            dst = torch.zeros(4096 * world, dtype=torch.uint8, device="cuda")
            device.allgather_dev(src.data_ptr(), dst.data_ptr(), 4096)
            # 2: rank r contributes (r + 1) * 1000 bytes
            sizes = [(r + 1) * 1000 for r in range(world)]
            offs = [sum(sizes[:r]) for r in range(world)]
            src2 = mark(rank, 1, sizes[rank])
            dst2 = torch.zeros(sum(sizes), dtype=torch.uint8, device="cuda")
            device.allgatherv_dev(src2.data_ptr(), dst2.data_ptr(), sizes, offs)
            # 3: rank a sends (a * world + b + 1) * 100 bytes to rank b
            sb = [(rank * world + b + 1) * 100 for b in range(world)]
            rb = [(a * world + rank + 1) * 100 for a in range(world)]
            src3 = torch.cat([mark(rank, 2 + b, sb[b]) for b in range(world)])
            dst3 = torch.zeros(sum(rb), dtype=torch.uint8, device="cuda")
            device.alltoallv_dev(src3.data_ptr(), sb, dst3.data_ptr(), rb)
            device.stream_wait(timeout_ms)
            want1 = torch.cat([mark(r, 0, 4096) for r in range(world)])
            want2 = torch.cat([mark(r, 1, sizes[r]) for r in range(world)])
            want3 = torch.cat([mark(a, 2 + rank, rb[a]) for a in range(world)])
            side.synchronize()
            for k, (got, want) in enumerate(((dst, want1), (dst2, want2), (dst3, want3)), 1):
                if not torch.equal(got, want):
                    return "self-test %d of the library's communicator: wrong bytes arrived" % k
        return None
    except Exception as err:                                  # noqa: B902 — whatever it is, the other route is there
        return "%s: %s" % (type(err).__name__, err)
    finally:
        device.use_torch_stream()                             # back to the stream the caller works on


def drop_abi_comm():
    if _abi["dev"] is not None:
        _abi["dev"].comm_destroy()
    _abi.update(dev=None, rank=0, world=1, counts=None)


def abi_comm():
    """The device of use_abi_comm(), or None."""
    return _abi["dev"]


def _abi_for(t):
    return _abi["dev"] if (_abi["dev"] is not None and t.is_cuda) else None


def group_of_one_exchanges():
    """SNPGPU_DIST_AT_WORLD_1=1 (functional tests on a one-GPU box): a process group of ONE rank still makes every collective
    call of the N > 1 path, so that the RCCL entry points themselves execute — on device tensors, no host staging — where only
    one GPU is there to run them.  Not a measurement mode."""
    return os.environ.get("SNPGPU_DIST_AT_WORLD_1") == "1"


def _alone():
    """Nothing to exchange: no process group, or a group of one (unless the test hook above asks for the calls anyway)."""
    if _abi["dev"] is not None:
        return _abi["world"] == 1 and not group_of_one_exchanges()
    if not dist.is_initialized():
        return True
    return dist.get_world_size() == 1 and not group_of_one_exchanges()


def _world():
    return _abi["world"] if _abi["dev"] is not None else dist.get_world_size()


def shard_bounds(n_items, rank, world):
    """Contiguous block of ceil(n/world) items for `rank` (sorted-dir order is kept inside and across ranks)."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def _via_host(t):
    """gloo moves host tensors only: the single-GPU functional tests of the N > 1 path stage through the host."""
    return dist.is_initialized() and dist.get_backend() == "gloo" and t.is_cuda


def all_gather_varlen(t):
    """All-gather of 1-D tensors whose lengths differ per rank.  Returns (concatenation in rank order, lengths)."""
    if _alone():
        return t, [int(t.numel())]
    dev = _abi_for(t)
    if dev is not None:
        # the lengths first (one word per rank through the same library; the host needs them to size the result), then every
        # rank's block straight to its place: no padding, no concatenation
        world = _abi["world"]
        if _abi["counts"] is None:
            _abi["counts"] = (torch.zeros(1, dtype=torch.int64, device=t.device), torch.zeros(world, dtype=torch.int64, device=t.device),
                              torch.zeros(1, dtype=torch.int64).pin_memory(), torch.zeros(world, dtype=torch.int64).pin_memory())
        d_n, d_counts, h_n, h_counts = _abi["counts"]
        h_n[0] = t.numel()
        d_n.copy_(h_n, non_blocking=True)
        dev.allgather_dev(d_n.data_ptr(), d_counts.data_ptr(), 8)
        h_counts.copy_(d_counts, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        counts = [int(c) for c in h_counts.tolist()]
        item = t.element_size()
        out = torch.empty(sum(counts), dtype=t.dtype, device=t.device)
        offs = [0]
        for c in counts[:-1]:
            offs.append(offs[-1] + c * item)
        tc = t.contiguous()
        dev.allgatherv_dev(tc.data_ptr() if tc.numel() else 0, out.data_ptr() if out.numel() else tc.data_ptr(), [c * item for c in counts], offs)
        return out, counts
    if _via_host(t):
        out, counts = all_gather_varlen(t.cpu())
        return out.to(t.device), counts
    world = dist.get_world_size()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    counts = torch.zeros(world, dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(counts, n)
    counts = [int(c) for c in counts.tolist()]
    cap = max(counts) if counts else 0
    if cap == 0:
        return t[:0], counts
    padded = torch.zeros(cap, dtype=t.dtype, device=t.device)
    padded[:t.numel()] = t
    out = torch.empty(world * cap, dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * cap:r * cap + counts[r]] for r in range(world)]), counts


def all_gather_rows(rows, n_total):
    """rows: this rank's (n_local, row_bytes) uint8 block of the packed matrix, blocks as in shard_bounds.
    Returns the (n_total, row_bytes) matrix on every rank."""
    if _alone():
        return rows
    if _abi_for(rows) is not None:
        out = torch.zeros((n_total, rows.shape[1]), dtype=rows.dtype, device=rows.device)
        return all_gather_rows_into(rows, n_total, out)[:n_total]
    if _via_host(rows):
        return all_gather_rows(rows.cpu(), n_total).to(rows.device)
    world = dist.get_world_size()
    per = (n_total + world - 1) // world
    width = rows.shape[1]
    padded = torch.zeros((per, width), dtype=rows.dtype, device=rows.device)
    padded[:rows.shape[0]] = rows
    out = torch.empty((world * per, width), dtype=rows.dtype, device=rows.device)
    dist.all_gather_into_tensor(out, padded)
    return out[:n_total]


def all_gather_rows_into(rows, n_total, out):
    """As all_gather_rows, but straight into the first rows of `out` (at least max(n_total, world * ceil(n_total / world))
    rows; the distance kernel wants the matrix padded to whole 128-row tiles, the padding rows stay zero)."""
    if _alone():
        out[:rows.shape[0]].copy_(rows)
        return out
    dev = _abi_for(rows)
    if dev is not None:
        # every rank's block of rows straight to its place in `out` (rows past n_total are not written: they stay what they were)
        world = _abi["world"]
        width = rows.shape[1] * rows.element_size()
        bounds = [shard_bounds(n_total, r, world) for r in range(world)]
        rc = rows.contiguous()
        dev.allgatherv_dev(rc.data_ptr() if rc.numel() else 0, out.data_ptr(), [(hi - lo) * width for lo, hi in bounds], [lo * width for lo, _ in bounds])
        return out
    world = dist.get_world_size()
    per = (n_total + world - 1) // world
    if _via_host(rows):
        out[:n_total].copy_(all_gather_rows(rows.cpu(), n_total).to(rows.device))
        return out
    if rows.shape[0] == per:
        padded = rows
    else:
        padded = torch.zeros((per, rows.shape[1]), dtype=rows.dtype, device=rows.device)
        padded[:rows.shape[0]] = rows
    dist.all_gather_into_tensor(out[:world * per], padded)
    if world * per > n_total:
        out[n_total:world * per].zero_()
    return out


def upper_tiles(n):
    """[(bi, bj)] of the upper triangle in the kernel's enumeration order (row-major, bj >= bi)."""
    nt = (n + DIST_TILE - 1) // DIST_TILE
    return [(bi, bj) for bi in range(nt) for bj in range(bi, nt)]


def tiles_of_rank(n, rank, world):
    return [t for i, t in enumerate(upper_tiles(n)) if i % world == rank]


def sum_partial_distances(partial):
    """Every rank filled only its own tiles (and their mirror images) of an n x n int32 matrix of zeros.  The sum goes through
    torch.distributed; a host that drives the library's communicator without a process group has the row-band
    exchange (RowBands.exchange) for this and must not get a partial matrix back in silence."""
    if _alone():
        return partial
    if not dist.is_initialized():
        raise RuntimeError("sum_partial_distances needs torch.distributed (the library's communicator has no all-reduce): "
                           "use RowBands.exchange, which moves every tile once to the owner of its row")
    dist.all_reduce(partial, op=dist.ReduceOp.SUM)
    return partial


class RowBands(object):
    """Who computes which 128 x 128 tile, who owns which rows, and what travels where — a pure function of (n, world), so
    every rank derives the whole exchange plan itself and no metadata is ever sent.

    Tiles: upper triangle, tile t (row-major enumeration) computed by rank t % world, together with its mirror image.
    Rows: tile rows in contiguous bands, band q = shard_bounds(n_tile_rows, q, world)."""

    def __init__(self, n, world):
        self.n, self.world = n, world
        self.nt = (n + DIST_TILE - 1) // DIST_TILE
        self.n_padded = self.nt * DIST_TILE
        self.bands = [shard_bounds(self.nt, q, world) for q in range(world)]
        owner_of_row = [0] * self.nt
        for q, (lo, hi) in enumerate(self.bands):
            for r in range(lo, hi):
                owner_of_row[r] = q
        # blocks[src][dst] = [(tile_row, tile_col)] of the full matrix that src computed and dst owns, in a fixed order
        self._index = {}
        self.blocks = [[[] for _ in range(world)] for _ in range(world)]
        for t, (bi, bj) in enumerate(upper_tiles(n)):
            src = t % world
            self.blocks[src][owner_of_row[bi]].append((bi, bj))
            if bi != bj:
                self.blocks[src][owner_of_row[bj]].append((bj, bi))

    def band_rows(self, rank):
        """[lo, hi) matrix rows of the band of `rank` (clipped to n)."""
        lo, hi = self.bands[rank]
        return min(self.n, lo * DIST_TILE), min(self.n, hi * DIST_TILE)

    def exchange(self, partial, rank):
        """partial: this rank's (n_padded, n_padded) int32 matrix holding the tiles it computed (and their mirror images).
        Returns the (band tile rows * 128, n_padded) band of complete rows this rank owns.

        ALIASING: over the library's communicator the returned band is this object's own buffer (made once per rank and device, so
        that a step allocates nothing) and the NEXT exchange() of the same RowBands overwrites it; with one rank it is a view of
        `partial`.  A caller that keeps band k while band k + 1 is computed must .clone() it; the torch.distributed route returns a
        fresh tensor every time."""
        nt, T, world = self.nt, DIST_TILE, self.world
        lo, hi = self.bands[rank]
        if world == 1 and not group_of_one_exchanges():                   # one rank owns every row: the matrix it computed is the band
            return partial[:(hi - lo) * T]
        dev_abi = _abi_for(partial)
        if dev_abi is not None:
            return self._exchange_abi(dev_abi, partial, rank)
        band = torch.zeros(((hi - lo) * T, self.n_padded), dtype=partial.dtype, device=partial.device)
        p4 = partial.view(nt, T, nt, T)
        b4 = band.view(max(hi - lo, 0), T, nt, T) if hi > lo else None
        in_splits = [len(self.blocks[rank][dst]) for dst in range(world)]
        out_splits = [len(self.blocks[src][rank]) for src in range(world)]
        dev = partial.device
        plan = self._index.get((rank, str(dev)))
        if plan is None:                                                  # index tensors of this rank's plan, built once
            send_list = [blk for dst in range(world) for blk in self.blocks[rank][dst]]
            recv_list = [blk for src in range(world) for blk in self.blocks[src][rank]]
            plan = tuple(torch.tensor(v, dtype=torch.int64, device=dev) for v in (
                [b[0] for b in send_list], [b[1] for b in send_list], [b[0] - lo for b in recv_list], [b[1] for b in recv_list]))
            self._index[(rank, str(dev))] = plan
        rows, cols, rrows, rcols = plan
        if rows.numel():
            send = p4[rows, :, cols, :].contiguous()                      # (k, 128, 128)
        else:
            send = torch.zeros((0, T, T), dtype=partial.dtype, device=dev)
        if _alone():
            recv = send
        else:
            recv = torch.empty((sum(out_splits), T, T), dtype=partial.dtype, device=dev)
            if _via_host(send):
                recv_h = recv.cpu()
                dist.all_to_all_single(recv_h, send.cpu(), out_splits, in_splits)
                recv = recv_h.to(dev)
            else:
                dist.all_to_all_single(recv, send, out_splits, in_splits)
        if rrows.numel():
            b4[rrows, :, rcols, :] = recv
        return band

    def _exchange_abi(self, dev, partial, rank):
        """The same exchange as library calls on the context's stream: k_tiles_gather packs this rank's tiles in destination order,
        one group of ncclSend / ncclRecv moves them, k_tiles_scatter puts what arrived into the band.  The buffers and the index
        arrays are made once per (rank, device); a band is written completely every time (every tile of its rows arrives exactly
        once), so it needs no clearing.  One rank alone owns every row: the matrix it computed is the band."""
        nt, T, world = self.nt, DIST_TILE, self.world
        lo, hi = self.bands[rank]
        key = ("abi", rank, str(partial.device))
        plan = self._index.get(key)
        if plan is None:
            send_list = [blk for dst in range(world) for blk in self.blocks[rank][dst]]
            recv_list = [blk for src in range(world) for blk in self.blocks[src][rank]]
            as_u32 = lambda v: torch.tensor(v, dtype=torch.int32, device=partial.device)     # noqa: E731  (tile indices: far below 2^31)
            plan = {
                "s_rows": as_u32([b[0] for b in send_list]), "s_cols": as_u32([b[1] for b in send_list]),
                "r_rows": as_u32([b[0] - lo for b in recv_list]), "r_cols": as_u32([b[1] for b in recv_list]),
                "n_send": len(send_list), "n_recv": len(recv_list),
                "send": torch.empty((max(len(send_list), 1), T, T), dtype=torch.int32, device=partial.device),
                "recv": torch.empty((max(len(recv_list), 1), T, T), dtype=torch.int32, device=partial.device),
                "band": torch.zeros((max(hi - lo, 0) * T, self.n_padded), dtype=torch.int32, device=partial.device),
                "send_bytes": [len(self.blocks[rank][dst]) * T * T * 4 for dst in range(world)],
                "recv_bytes": [len(self.blocks[src][rank]) * T * T * 4 for src in range(world)],
            }
            self._index[key] = plan
        if partial.dtype != torch.int32 or not partial.is_contiguous():
            raise ValueError("the distance matrix must be a contiguous int32 tensor")
        dev.tiles_gather_dev(partial.data_ptr(), self.n_padded, plan["s_rows"].data_ptr(), plan["s_cols"].data_ptr(), plan["n_send"], plan["send"].data_ptr())
        dev.alltoallv_dev(plan["send"].data_ptr(), plan["send_bytes"], plan["recv"].data_ptr(), plan["recv_bytes"])
        if hi > lo:
            dev.tiles_scatter_dev(plan["recv"].data_ptr(), plan["r_rows"].data_ptr(), plan["r_cols"].data_ptr(), plan["n_recv"], plan["band"].data_ptr(),
                                  self.n_padded)
        return plan["band"]

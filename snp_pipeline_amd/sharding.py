"""Sample sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests).

The path shards by samples: each rank parses its own VCFs and scans its own pileups with no
communication.  There are exactly two exchange steps, both all-gathers:
  C1  the per-rank SNP site keys (variable length) -> every rank runs the same merge -> identical global snplist
  C2  the per-rank rows of the 4-bit packed consensus matrix -> every rank holds the full matrix
The N x N distance is then dealt out in 128 x 128 tiles of the upper triangle, tile t to rank t % world, and the
partial matrices are summed (each entry is written by exactly one rank).
"""
import torch
import torch.distributed as dist

DIST_TILE = 128          # csrc/distance.hip


def shard_bounds(n_items, rank, world):
    """Contiguous block of ceil(n/world) items for `rank` (sorted-dir order is kept inside and across ranks)."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def all_gather_varlen(t):
    """All-gather of 1-D tensors whose lengths differ per rank.  Returns (concatenation in rank order, lengths)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t, [int(t.numel())]
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
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rows
    world = dist.get_world_size()
    per = (n_total + world - 1) // world
    width = rows.shape[1]
    padded = torch.zeros((per, width), dtype=rows.dtype, device=rows.device)
    padded[:rows.shape[0]] = rows
    out = torch.empty((world * per, width), dtype=rows.dtype, device=rows.device)
    dist.all_gather_into_tensor(out, padded)
    return out[:n_total]


def upper_tiles(n):
    """[(bi, bj)] of the upper triangle in the kernel's enumeration order (row-major, bj >= bi)."""
    nt = (n + DIST_TILE - 1) // DIST_TILE
    return [(bi, bj) for bi in range(nt) for bj in range(bi, nt)]


def tiles_of_rank(n, rank, world):
    return [t for i, t in enumerate(upper_tiles(n)) if i % world == rank]


def sum_partial_distances(partial):
    """Every rank filled only its own tiles (and their mirror images) of an n x n int32 matrix of zeros."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(partial, op=dist.ReduceOp.SUM)
    return partial

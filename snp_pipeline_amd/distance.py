"""distance subcommand: all-pairs SNP distances from snpma.fasta.

Host mirror of snppipeline/distance.py:14-118.  The pair counting (utils.calculate_sequence_distance for every pair)
runs in the HIP distance kernel (csrc/distance.hip) behind ``Device.distance``; this file parses the multi-FASTA,
orders the ids and writes the two TSV layouts.
"""
from __future__ import print_function

import numpy as np

from . import snp_matrix
from . import utils


def distance_matrix(dev, seqs):
    """seqs: {id: sequence}.  Returns (sorted ids, (n, n) int32 matrix).

    Unequal lengths behave as in the reference: utils.calculate_sequence_distance (utils.py:1156-1158) walks
    range(len(seq1)) for every pair of itertools.combinations(sorted ids) — a longer second sequence is simply cut, a shorter
    one raises IndexError at seq2[pos].  So the lengths must be non-decreasing in sorted-id order, and a pair is compared over
    the length of its first sequence: the shorter rows are padded with '-' (never counted) and the kernel does the rest."""
    ids = sorted(seqs.keys())
    n = len(ids)
    if n == 0:
        return ids, np.zeros((0, 0), dtype=np.int32)
    lens = [len(seqs[i]) for i in ids]
    for a, b in zip(lens, lens[1:]):
        if b < a:
            raise IndexError("string index out of range")                     # what seq2[pos] raises in utils.py:1158
    s = lens[-1]
    if s == 0:
        return ids, np.zeros((n, n), dtype=np.int32)
    sym = np.full((n, s), 0x2D, dtype=np.uint8)
    for r, i in enumerate(ids):
        sym[r, :lens[r]] = np.frombuffer(seqs[i].encode("latin-1"), dtype=np.uint8)
    return ids, dev.distance(sym)


def distance_of_matrix(dev, file_ids, sym, lens):
    """distance_matrix for the arrays of snp_matrix.load_matrix (no Python strings in between): equal ids keep their last
    record (the reference's dict, distance.py:80-84), rows go into sorted-id order, lengths follow the same rule."""
    last = {}
    for r, i in enumerate(file_ids):
        last[i] = r
    ids = sorted(last)
    rows = np.asarray([last[i] for i in ids], dtype=np.int64)
    n = len(ids)
    if n == 0:
        return ids, np.zeros((0, 0), dtype=np.int32)
    ordered = lens[rows]
    if (ordered[1:] < ordered[:-1]).any():
        raise IndexError("string index out of range")                         # what seq2[pos] raises in utils.py:1158
    if sym.shape[1] == 0:
        return ids, np.zeros((n, n), dtype=np.int32)
    if not np.array_equal(rows, np.arange(len(file_ids))):
        sym = sym[rows]                                                         # (snp_matrix writes sorted sample order: usually a no-op)
    return ids, dev.distance(np.ascontiguousarray(sym))


def _write_tsv(path, layout, ids, mat):
    """distance.py:100-114 through the library's host formatter (csrc/tsv_out.hip): at 10 000 samples the pairwise file has
    10^8 lines, ~35 s of Python string formatting for 38 ms of kernel."""
    from . import _lib as L
    names = [i.encode("utf-8") for i in ids]
    off = np.zeros(len(names) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in names], out=off[1:])
    blob = b"".join(names)
    m = np.ascontiguousarray(mat, dtype=np.int32)
    rc = L.load().snpgpu_write_distance_tsv(path.encode(), layout, blob, off.ctypes.data, len(names), m.ctypes.data, m.shape[1] if m.ndim == 2 else 0)
    if rc == L.E_IO:
        raise IOError("cannot write %s" % path)
    if rc != 0:
        raise RuntimeError("snpgpu_write_distance_tsv failed (%d)" % rc)


def write_pairwise(path, ids, mat):
    _write_tsv(path, 0, ids, mat)


def write_matrix(path, ids, mat):
    _write_tsv(path, 1, ids, mat)


def calculate_snp_distances(args):
    """Entry point of ``cfsan_snp_pipeline distance`` (cfsan_snp_pipeline.py:448-457)."""
    utils.print_log_header()
    utils.print_arguments(args)

    input_file = args.inputFile
    pairwise_file = args.pairwiseFile
    matrix_file = args.matrixFile
    if utils.verify_existing_input_files("SNP matrix file", [input_file]) > 0:
        utils.global_error("Error: cannot calculate sequence distances without the snp matrix file.")
    if not pairwise_file and not matrix_file:
        utils.global_error("Error: no output file specified.")

    rebuild_pairwise = pairwise_file and utils.target_needs_rebuild([input_file], pairwise_file)
    rebuild_matrix = matrix_file and utils.target_needs_rebuild([input_file], matrix_file)
    if not (args.forceFlag or rebuild_pairwise or rebuild_matrix):
        utils.verbose_print("Distance files have already been freshly built.  Use the -f option to force a rebuild.")
        return

    file_ids, sym, lens = snp_matrix.load_matrix(input_file)
    utils.verbose_print("# %s %s" % (utils.timestamp(), "Calculating all pairwise distances"))
    from .device import default_device
    ids, mat = distance_of_matrix(default_device(), file_ids, sym, lens)
    if pairwise_file:
        write_pairwise(pairwise_file, ids, mat)
    if matrix_file:
        write_matrix(matrix_file, ids, mat)

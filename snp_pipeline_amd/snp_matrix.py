"""snp_matrix subcommand: concatenate the per-sample consensus FASTA files into snpma.fasta.

Host side of snppipeline/snp_matrix.py:13-119 (a byte copy in sorted sample-directory order; no arithmetic).
``create_snp_matrix`` is the CLI contract of the step: its input checks, their order, the messages and the freshness test are
those of snp_matrix.py:68-110, said the same way; the copy (whole files, bytes) is this build's own.
``read_matrix`` additionally returns the sequences as a samples x sites byte matrix for the distance kernel.
"""
from __future__ import print_function

import os

import numpy as np

from . import utils


def create_snp_matrix(args):
    """Entry point of ``cfsan_snp_pipeline snp_matrix`` (cfsan_snp_pipeline.py:429-443)."""
    utils.print_log_header()
    utils.print_arguments(args)

    sample_directories_list_filename = args.sampleDirsFile
    if utils.verify_non_empty_input_files("File of sample directories", [sample_directories_list_filename]) > 0:
        utils.global_error(None)
    with open(sample_directories_list_filename, "r") as f:
        dirs = [line.rstrip() for line in f]
    dirs = sorted(d for d in dirs if d)

    consensus_files = []
    bad_file_count = 0
    for sample_directory in dirs:
        path = os.path.join(sample_directory, args.consFileName)
        if utils.verify_non_empty_input_files("Consensus fasta file", [path]) == 1:
            bad_file_count += 1
        else:
            consensus_files.append(path)
    if bad_file_count == len(dirs):
        utils.global_error("Error: all %d consensus fasta files were missing or empty." % bad_file_count)
    elif bad_file_count > 0:
        utils.sample_error("Error: %d consensus fasta files were missing or empty." % bad_file_count, continue_possible=True)

    snpma_file_path = args.snpmaFile
    if not args.forceFlag and not utils.target_needs_rebuild(consensus_files, snpma_file_path):
        utils.verbose_print("SNP matrix %s has already been freshly built.  Use the -f option to force a rebuild." % snpma_file_path)
        return

    # snp_matrix.py:112-117 copies the files line by line in text mode: the bytes, with "\r\n" and lone "\r" turned into
    # "\n" (universal newlines) and an error for text that is not valid UTF-8.  Whole files at a time here (10 000 files
    # of 200 kB = 3.3e7 lines at configs[4]).
    with open(snpma_file_path, "wb") as output_file:
        for path in consensus_files:
            utils.verbose_print("Merging " + path)
            with open(path, "rb") as input_file:
                data = input_file.read()
            if not data.isascii():
                data.decode("utf-8")                            # raises UnicodeDecodeError where the reference's read does
            if b"\r" in data:
                data = data.replace(b"\r\n", b"\n").replace(b"\r", b"\n")
            output_file.write(data)


def read_matrix(path):
    """Parse a multi-FASTA the way distance.py:76-84 does.  Returns (ids in file order, {id: sequence str})."""
    seqs = {}
    curr = None
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(">"):
                curr = line.lstrip(">")
                seqs[curr] = []
            else:
                if curr is None:                    # distance.py:84 reads curr_sample before any header has set it
                    raise UnboundLocalError("local variable 'curr_sample' referenced before assignment")
                seqs[curr].append(line)
    return {k: "".join(v) for k, v in seqs.items()}


def load_matrix(path):
    """The same parse straight into a byte matrix, by the library's host code (csrc/fasta_in.hip): returns (ids in file order,
    (records x longest) uint8 matrix padded with '-', lengths).  Raises UnboundLocalError for sequence text before the first
    header, like read_matrix and the reference (golden: distance_runs).  Duplicate ids are all returned (the reference's dict keeps the last one: see distance.py)."""
    import ctypes as C
    from . import _lib as L
    lib = L.load()
    n, longest, names_bytes = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = lib.snpgpu_fasta_scan(os.fsencode(path), C.byref(n), C.byref(longest), C.byref(names_bytes))
    if rc == L.E_IO:
        raise IOError("cannot read %s" % path)
    if rc == L.E_UNSUPPORTED:
        raise UnboundLocalError("local variable 'curr_sample' referenced before assignment")
    if rc != 0:
        raise RuntimeError("snpgpu_fasta_scan failed (%d)" % rc)
    mat = np.empty((n.value, longest.value), dtype=np.uint8)
    lens = np.zeros(n.value, dtype=np.uint64)
    names = C.create_string_buffer(max(1, names_bytes.value))
    off = np.zeros(n.value + 1, dtype=np.uint64)
    rc = lib.snpgpu_fasta_load(os.fsencode(path), n.value, longest.value, 0x2D, mat.ctypes.data if mat.size else None, lens.ctypes.data,
                               names, off.ctypes.data)
    if rc != 0:
        raise IOError("%s changed while it was read" % path)
    raw = names.raw
    ids = [raw[int(off[i]):int(off[i + 1])].decode("utf-8") for i in range(n.value)]
    return ids, mat, lens

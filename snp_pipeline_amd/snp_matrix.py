"""snp_matrix subcommand: concatenate the per-sample consensus FASTA files into snpma.fasta.

Host mirror of snppipeline/snp_matrix.py:13-119 (a byte copy in sorted sample-directory order; no arithmetic).
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

    with open(snpma_file_path, "w") as output_file:
        for path in consensus_files:
            utils.verbose_print("Merging " + path)
            with open(path, "r") as input_file:
                for line in input_file:
                    output_file.write(line)


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
                seqs[curr].append(line)             # KeyError(None) if data precedes the first header, like the reference's NameError
    return {k: "".join(v) for k, v in seqs.items()}

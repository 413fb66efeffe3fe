"""snp_reference subcommand: the reference bases at the snplist positions as a FASTA file.

Host side of snppipeline/snp_reference.py:12-77 and utils.write_reference_snp_file (utils.py:1091-1110).  There is
no arithmetic in this step — one byte gathered per site — so it stays on the host; it is here because
referenceSNP.fasta is one of the top-level outputs the regression suite diffs.

``create_snp_reference_seq`` is the CLI contract of the step — which inputs are checked, in which order, with which messages, and
the freshness test — and therefore says the same things in the same order as snp_reference.py:50-77 does; the gather itself
(``_gather_plain``, ``read_fasta_sequences``) is this build's own.
"""
from __future__ import absolute_import

from . import utils
from .utils import verbose_print


def read_fasta_sequences(path):
    """{record id: sequence} the way Bio.SeqIO.to_dict(SeqIO.parse(path, "fasta")) sees the file: the id is the first
    word of the header, the sequence is the data lines with all whitespace removed; a repeated id is an error."""
    fast = utils.fasta_records_ascii(path)
    if fast is not None:
        seqs = {}
        for name, seq in fast:
            if name in seqs:
                raise ValueError("Duplicate key '%s'" % name)          # what SeqIO.to_dict raises
            seqs[name] = seq.decode("ascii")
        return seqs
    seqs = {}
    name, parts = None, []
    with open(path, "r") as f:
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    seqs[name] = "".join(parts)
                words = line[1:].split()
                name, parts = (words[0] if words else ""), []
                if name in seqs:
                    raise ValueError("Duplicate key '%s'" % name)      # what SeqIO.to_dict raises
            elif name is not None:
                parts.append("".join(line.split()))
    if name is not None:
        seqs[name] = "".join(parts)
    return seqs


def _gather_plain(match_dict, snp_list_file_path, snp_reference_file_path):
    """The same file from arrays, for the plain case: a snplist the library's reader takes (utils.read_snp_position_arrays) and
    ASCII sequences — one numpy gather per contig instead of a Python loop per site.  Returns False when the case is not plain."""
    import numpy as np
    try:
        names, cidx, pos = utils.read_snp_position_arrays(snp_list_file_path)
    except ValueError:
        return False                                           # the loop below raises where (and only where) the reference does
    if any(not seq.isascii() for seq in match_dict.values()):
        return False
    by_name = {n: i for i, n in enumerate(names)}
    with open(snp_reference_file_path, "w") as out:
        for ordered_id in sorted(match_dict.keys()):
            seq = np.frombuffer(match_dict[ordered_id].encode("ascii"), dtype=np.uint8)
            if ordered_id in by_name:
                p = pos[cidx == by_name[ordered_id]] - 1       # python indexing: 0 reads the last base, past the end raises
                picked = np.take(seq, p)                        # (mode "raise": IndexError like str indexing, negatives wrap)
                picked = np.where((picked >= 97) & (picked <= 122), picked - 32, picked).astype(np.uint8)
                ref_str = picked.tobytes().decode("ascii")
            else:
                ref_str = ""
            utils.write_fasta_record(out, ordered_id, ref_str)
    return True


def write_reference_snp_file(reference_file_path, snp_list_file_path, snp_reference_file_path, match_dict=None):
    """utils.py:1091-1110: for every contig of the reference in sorted id order, the upper-cased reference bases at
    the snplist positions of that contig, in snplist order (python indexing: position 0 reads the last base, a
    position past the end raises IndexError).  match_dict: the reference's sequences when the caller has read them already."""
    if match_dict is None:
        match_dict = read_fasta_sequences(reference_file_path)
    if _gather_plain(match_dict, snp_list_file_path, snp_reference_file_path):
        return
    with open(snp_list_file_path, "r") as snp_list_file:
        position_list = [line.split(None, 2)[0:2] for line in snp_list_file]
    by_contig = {}
    for item in position_list:
        chrom_id, pos = item                                  # a line with fewer than two fields raises, as in the reference
        by_contig.setdefault(chrom_id, []).append(pos)
    with open(snp_reference_file_path, "w") as out:
        for ordered_id in sorted(match_dict.keys()):
            seq = match_dict[ordered_id]
            ref_str = "".join(seq[int(pos) - 1].upper() for pos in by_contig.get(ordered_id, ()))
            utils.write_fasta_record(out, ordered_id, ref_str)


def create_snp_reference_seq(args):
    """args: referenceFile, snpListFile, snpRefFile, forceFlag (snp_reference.py:12-77)."""
    utils.print_log_header()
    utils.print_arguments(args)
    # the step's inputs in the order the reference checks them: (how, label in the log, path, what the step says when it is bad)
    inputs = ((utils.verify_existing_input_files, "Snplist file", args.snpListFile, "snplist file"),
              (utils.verify_non_empty_input_files, "Reference file", args.referenceFile, "reference fasta file"))
    for verify, label, path, what in inputs:
        if verify(label, [path]) > 0:
            utils.global_error("Error: cannot create the snp reference sequence without the %s." % what)
    target = args.snpRefFile
    if not args.forceFlag and not utils.target_needs_rebuild([args.referenceFile, args.snpListFile], target):
        verbose_print("SNP reference sequence %s has already been freshly built.  Use the -f option to force a rebuild." % target)
        return
    write_reference_snp_file(args.referenceFile, args.snpListFile, target)

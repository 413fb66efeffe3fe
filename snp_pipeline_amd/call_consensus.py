"""call_consensus subcommand: consensus base of one sample at every snplist position.

Host mirror of snppipeline/call_consensus.py:18-192.  Argument handling, input checks, freshness and the FASTA/VCF
writers live here; reading the pileup, the per-position base counting and the caller's filters run in the HIP
kernels (csrc/scan.hip, csrc/consensus.hip) behind ``Device.call_consensus``.
"""
from __future__ import print_function

import mmap
import os

from . import _lib as L
from . import device as devmod
from . import utils
from . import vcf_writer


def consensus_for_sample(dev, pileup_bytes, snp_list, excluded_positions, params, want_counts=False, extra_positions=None):
    """snp_list: [(chrom str, pos int)] in snplist order; excluded_positions: set of the same.
    Returns (consensus str in snplist order, SiteSet, ConsensusResult)."""
    keys = [(c.encode(), p) for c, p in snp_list]
    snps = set(keys)
    excl = {(c.encode(), p) for c, p in excluded_positions}
    extra = [k for k in sorted(excl) if k not in snps]
    all_keys = keys + extra
    flags = [(L.SITE_IN_SNPLIST if k in snps else 0) | (L.SITE_EXCLUDED if k in excl else 0) for k in all_keys]
    ss = dev.siteset(all_keys, flags)
    res = dev.call_consensus(ss, pileup_bytes, params, want_counts=want_counts)
    idx = ss.index_of[:len(keys)]
    consensus = bytes(int(res.bases[i]) if i >= 0 else 0x2D for i in idx).decode("ascii")
    return consensus, ss, res


def call_consensus(args):
    """Entry point of ``cfsan_snp_pipeline call_consensus`` (cfsan_snp_pipeline.py:345-410)."""
    utils.print_log_header()
    utils.print_arguments(args)

    snp_list_file_path = args.snpListFile
    all_pileup_file_path = args.allPileupFile
    sample_directory = os.path.dirname(os.path.abspath(all_pileup_file_path))
    sample_name = os.path.basename(sample_directory)
    consensus_file_path = args.consensusFile
    consensus_file_dir = os.path.dirname(os.path.abspath(consensus_file_path))
    vcf_file_name = args.vcfFileName
    vcf_file_path = os.path.join(consensus_file_dir, vcf_file_name) if vcf_file_name else None

    if utils.verify_existing_input_files("Snplist file", [snp_list_file_path]) > 0:
        utils.global_error("Error: cannot call consensus without the snplist file.")
    if utils.verify_non_empty_input_files("Pileup file", [all_pileup_file_path]) > 0:
        utils.sample_error("Error: cannot call consensus without the pileup file.", continue_possible=False)
    source_files = [snp_list_file_path, all_pileup_file_path]

    exclude_file_path = args.excludeFile
    if exclude_file_path:
        if utils.verify_existing_input_files("Exclude file", [exclude_file_path]) > 0:
            utils.sample_error("Error: cannot call consensus without the file of excluded positions.", continue_possible=False)
        excluded_positions = utils.convert_vcf_file_to_snp_set(exclude_file_path)
        source_files.append(exclude_file_path)
    else:
        excluded_positions = set()

    if not args.forceFlag and not utils.target_needs_rebuild(source_files, consensus_file_path):
        utils.verbose_print("Consensus call file %s has already been freshly built.  Use the -f option to force a rebuild." % consensus_file_path)
        return

    snp_list = utils.read_snp_position_list(snp_list_file_path)
    utils.verbose_print("snp position list length = %d" % len(snp_list))
    utils.verbose_print("excluded snps list length = %d" % len(excluded_positions))
    utils.verbose_print("total snp position list length = %d" % (len(snp_list) + len(excluded_positions)))

    if args.vcfAllPos and vcf_file_name:
        utils.global_error("Error: --vcfAllPos is not provided by the MI355X build (diagnostic option of the reference).")

    params = devmod.make_params(args.minBaseQual, args.minConsFreq, args.minConsDpth, args.minConsStrdDpth, args.minConsStrdBias)
    dev = devmod.default_device()
    with open(all_pileup_file_path, "rb") as f:
        with mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
            consensus, ss, res = consensus_for_sample(dev, mm, snp_list, excluded_positions, params, want_counts=True)
    in_snplist = (ss.flags & L.SITE_IN_SNPLIST) != 0
    utils.verbose_print("called consensus positions = %i" % int(((res.counts["status"] != L.ST_NO_LINE) & in_snplist).sum()))

    if vcf_file_name:
        vcf_writer.write_consensus_vcf(vcf_file_path, sample_name, args, ss, res, dev.line_offsets(ss))

    with open(consensus_file_path, "w") as fasta_file_object:
        utils.write_fasta_record(fasta_file_object, sample_name, consensus)

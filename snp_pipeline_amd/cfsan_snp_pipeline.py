#!/usr/bin/env python
"""``cfsan_snp_pipeline`` console entry for the post-alignment subcommands on MI355X.

Mirrors the argparse surface of snppipeline/cfsan_snp_pipeline.py for call_sites (:294-304), filter_regions (:309-324, list validation
:530-543), merge_sites (:329-340), call_consensus (:345-410), snp_matrix (:429-443), distance (:448-457) and snp_reference (:460-471): same
flags, defaults, type validators, per-subcommand exception hook, exit codes and "finished" banner, plus the
python-level helpers the reference's unit tests use (parse_command_line, parse_argument_list,
run_command_from_args ...).  Every other subcommand belongs to the alignment stage or to orchestration and is not
provided by this build: asking for one exits with an explanatory message.
"""
from __future__ import absolute_import

import argparse
import os
import sys

from . import call_consensus, call_sites, distance, filter_regions, hot_path, merge_sites, service, snp_matrix, snp_reference, utils
from .utils import __version__, verbose_print

NOT_PROVIDED = ("run", "data", "index_ref", "map_reads", "merge_vcfs",
                "collect_metrics", "combine_metrics", "purge")


class HelpParser(argparse.ArgumentParser):
    def error(self, message):
        sys.stderr.write("Error: %s\n" % message)
        sys.exit(2)


def _not_provided(args):
    utils.global_error("Error: the %s command is not part of the MI355X hot-path build; use the reference "
                       "cfsan_snp_pipeline for it (SNPGPU_REFERENCE_CLI=<its path>, or leave it further down PATH)." % args.subparser_name)


def reference_cli():
    """The reference's own console script for the subcommands outside the hot path (run, map_reads, merge_vcfs, collect_metrics
    ...): $SNPGPU_REFERENCE_CLI, or the next ``cfsan_snp_pipeline`` on PATH that is not this build's.  None when there is none."""
    here = os.path.realpath(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "cfsan_snp_pipeline"))
    me = os.path.realpath(sys.argv[0]) if sys.argv and sys.argv[0] else None
    given = os.environ.get("SNPGPU_REFERENCE_CLI")
    if given:                                                # (pointing it at this build would be a loop of execs)
        return given if os.access(given, os.X_OK) and os.path.realpath(given) not in (here, me) else None
    for d in os.environ.get("PATH", "").split(os.pathsep):
        cand = os.path.join(d or ".", "cfsan_snp_pipeline")
        if os.path.isfile(cand) and os.access(cand, os.X_OK) and os.path.realpath(cand) not in (here, me):
            return cand
    return None


def _min_cons_freq(value):
    fvalue = float(value)
    if fvalue <= 0.5 or fvalue > 1:
        raise argparse.ArgumentTypeError("Minimum consensus frequency must be > 0.5 and <= 1.0")
    return fvalue


def _min_cons_strand_bias(value):
    fvalue = float(value)
    if fvalue < 0.0 or fvalue > 0.5:
        raise argparse.ArgumentTypeError("Minimum consensus strand bias must be >= 0.0 and <= 0.5")
    return fvalue


def _common(sub):
    sub.add_argument("-v", "--verbose", dest="verbose", type=int, default=1, metavar="0..5", help="Verbose message level (0=no info, 5=lots)")
    sub.add_argument("--version", action="version", version="%(prog)s version " + __version__)


def parse_argument_list(argv):
    """Parse command line arguments.  argv: list of strings, the first one is the subcommand name."""
    fmt = argparse.ArgumentDefaultsHelpFormatter
    parser = HelpParser(description="Tools of the CFSAN SNP Pipeline hot path (MI355X build).", formatter_class=fmt)
    parser.add_argument("--version", action="version", version="%(prog)s version " + __version__)
    subparsers = parser.add_subparsers(dest="subparser_name", help=None, metavar="subcommand       ")
    subparsers.required = True

    sub = subparsers.add_parser("call_sites", help="Find the sites with high-confidence SNPs in a sample", formatter_class=fmt,
                                description="Find the sites with high-confidence SNPs in a sample.")
    sub.add_argument(dest="referenceFile", type=str, help="Relative or absolute path to the reference fasta file")
    sub.add_argument(dest="sampleDir", type=str, help="Relative or absolute directory of the sample")
    sub.add_argument("-f", "--force", dest="forceFlag", action="store_true", help="Force processing even when result files already exist and are newer than inputs")
    _common(sub)
    sub.set_defaults(func=call_sites.call_sites, excepthook=utils.handle_sample_exception)

    sub = subparsers.add_parser("call_sites_batch", help="call_sites for every sample directory, one process, all visible GPUs", formatter_class=fmt,
                                description="Extension of the MI355X build: the call_sites step for every sample directory listed in sampleDirsFile "
                                            "(samtools mpileup where the pileup is stale, then one streamed device call per GPU).")
    sub.add_argument(dest="referenceFile", type=str, help="Relative or absolute path to the reference fasta file")
    sub.add_argument(dest="sampleDirsFile", type=str, help="Relative or absolute path to file containing a list of directories -- one per sample")
    sub.add_argument("-f", "--force", dest="forceFlag", action="store_true", help="Force processing even when result files already exist and are newer than inputs")
    sub.add_argument("--siteCalling", dest="siteCalling", type=str, default=None, choices=call_sites.SITE_CALLING_MODES, metavar="MODE",
                     help="Who writes var.flt.vcf: varscan (the VarScan jar on CLASSPATH, as the reference), device (this build's restatement; parity unpinned), "
                          "existing (nobody: the files are inputs), auto (varscan when a jar is on CLASSPATH, else device).  Default: $SNPGPU_SITE_CALLING, else auto")
    _common(sub)
    sub.set_defaults(func=call_sites.call_sites_batch, excepthook=utils.handle_global_exception)

    sub = subparsers.add_parser("filter_regions", help="Remove abnormally dense SNPs from all samples", formatter_class=fmt,
                                description="Remove abnormally dense SNPs from the input VCF file, save the reserved SNPs into a new VCF file, and save the removed SNPs into another VCF file.")
    sub.add_argument(dest="sampleDirsFile", type=str, help="Relative or absolute path to file containing a list of directories -- one per sample")
    sub.add_argument(dest="refFastaFile", type=str, help="Relative or absolute path to the reference fasta file")
    sub.add_argument("-f", "--force", dest="forceFlag", action="store_true", help="Force processing even when result files already exist and are newer than inputs")
    sub.add_argument("-n", "--vcfname", dest="vcfFileName", type=str, default="var.flt.vcf", metavar="NAME", help="File name of the input VCF files which must exist in each of the sample directories")
    sub.add_argument("-l", "--edge_length", dest="edgeLength", type=int, default=500, metavar="EDGE_LENGTH", help="The length of the edge regions in a contig, in which all SNPs will be removed.")
    sub.add_argument("-w", "--window_size", dest="windowSizeList", type=int, default=[1000], nargs="*", metavar="WINDOW_SIZE", help="The length of the window in which the number of SNPs should be no more than max_num_snp.")
    sub.add_argument("-m", "--max_snp", dest="maxSnpsList", type=int, default=[3], nargs="*", metavar="MAX_NUM_SNPs", help="The maximum number of SNPs allowed in a window.")
    sub.add_argument("-g", "--out_group", dest="outGroupFile", type=str, default=None, metavar="OUT_GROUP", help="Relative or absolute path to the file indicating outgroup samples, one sample ID per line.")
    sub.add_argument("-M", "--mode", dest="mode", choices=["all", "each"], default="all", help="Control whether dense snp regions found in any sample are filtered from all of the samples, or each sample independently.")
    _common(sub)
    sub.set_defaults(func=filter_regions.filter_regions, excepthook=utils.handle_global_exception)

    sub = subparsers.add_parser("merge_sites", help="Prepare the list of sites having SNPs", formatter_class=fmt,
                                description="Combine the SNP positions across all samples into a single unified SNP list file identifying the positions and sample names where SNPs were called.")
    sub.add_argument(dest="sampleDirsFile", type=str, help="Relative or absolute path to file containing a list of directories -- one per sample")
    sub.add_argument(dest="filteredSampleDirsFile", type=str, help="Relative or absolute path to the output file that will be created containing the filtered list of sample directories -- one per sample.")
    sub.add_argument("-f", "--force", dest="forceFlag", action="store_true", help="Force processing even when result file already exists and is newer than inputs")
    sub.add_argument("-n", "--vcfname", dest="vcfFileName", type=str, default="var.flt.vcf", metavar="NAME", help="File name of the VCF files which must exist in each of the sample directories")
    sub.add_argument("--maxsnps", dest="maxSnps", type=int, default=-1, metavar="INT", help="Exclude samples having more than this maximum allowed number of SNPs. Set to -1 to disable this function.")
    sub.add_argument("-o", "--output", dest="snpListFile", type=str, default="snplist.txt", metavar="FILE", help="Output file.  Relative or absolute path to the SNP list file")
    _common(sub)
    sub.set_defaults(func=merge_sites.merge_sites, excepthook=utils.handle_global_exception)

    sub = subparsers.add_parser("call_consensus", help="Call the consensus base at high-confidence sites", formatter_class=fmt,
                                description="Call the consensus base for a sample at the specified positions where high-confidence SNPs were previously called in any of the samples.  Generates a single-sequence fasta file with one base per specified position.")
    sub.add_argument(dest="allPileupFile", type=str, help="Relative or absolute path to the genome-wide pileup file for this sample.")
    sub.add_argument("-f", "--force", dest="forceFlag", action="store_true", help="Force processing even when result file already exists and is newer than inputs.")
    sub.add_argument("-l", "--snpListFile", dest="snpListFile", type=str, default="snplist.txt", metavar="FILE", help="Relative or absolute path to the SNP list file across all samples.")
    sub.add_argument("-e", "--excludeFile", dest="excludeFile", type=str, default=None, metavar="FILE", help="VCF file of positions to exclude.")
    sub.add_argument("-o", "--output", dest="consensusFile", type=str, default="consensus.fasta", metavar="FILE", help="Output file. Relative or absolute path to the consensus fasta file for this sample.")
    sub.add_argument("-q", "--minBaseQual", dest="minBaseQual", type=int, default=0, metavar="INT", help="Mimimum base quality score to count a read.")
    sub.add_argument("-c", "--minConsFreq", dest="minConsFreq", type=_min_cons_freq, default=0.60, metavar="FREQ", help="Consensus frequency.")
    sub.add_argument("-D", "--minConsDpth", dest="minConsDpth", type=int, default=1, metavar="INT", help="Consensus depth.")
    sub.add_argument("-d", "--minConsStrdDpth", dest="minConsStrdDpth", type=int, default=0, metavar="INT", help="Consensus strand depth.")
    sub.add_argument("-b", "--minConsStrdBias", dest="minConsStrdBias", type=_min_cons_strand_bias, default=0, metavar="FREQ", help="Strand bias.")
    sub.add_argument("--vcfFileName", dest="vcfFileName", type=str, default=None, metavar="NAME", help="VCF Output file name.")
    sub.add_argument("--vcfRefName", dest="vcfRefName", type=str, default="Unknown reference", metavar="NAME", help="Name of the reference file.  This is only used in the generated VCF file header.")
    sub.add_argument("--vcfAllPos", dest="vcfAllPos", action="store_true", help="Flag to cause VCF file generation at all positions, not just the snp positions.")
    sub.add_argument("--vcfPreserveRefCase", dest="vcfPreserveRefCase", action="store_true", help="Emit each reference base in uppercase/lowercase as it appears in the reference sequence file.")
    sub.add_argument("--vcfFailedSnpGt", dest="vcfFailedSnpGt", type=str, default=".", choices=[".", "0", "1"], help="Controls the VCF file GT data element when a snp fails filters.")
    sub.add_argument("--amdMetricsRefFasta", dest="amdMetricsRefFasta", type=str, default=None, metavar="FILE", help="Extension of this build: reference fasta file; when given, the mean pileup depth (a by-product of the pileup scan) and the number of missing positions are recorded in the sample's metrics file, where the reference's collect_metrics reuses them instead of reading the pileup again.")
    sub.add_argument("--amdMetricsFile", dest="amdMetricsFile", type=str, default=None, metavar="FILE", help="Extension of this build: the metrics file to update (default: metrics in the sample directory).")
    _common(sub)
    sub.set_defaults(func=call_consensus.call_consensus, excepthook=utils.handle_sample_exception)

    # Extension of this build (no reference counterpart): call_consensus for all samples of a sampleDirsFile in one process
    sub = subparsers.add_parser("call_consensus_batch", help="call_consensus for every sample directory, one process, all visible GPUs", formatter_class=fmt,
                                description="Run the call_consensus step for every sample directory listed in sampleDirsFile in one process: the pileup files are streamed through the visible GPUs, samples dealt round-robin.  Options as call_consensus; file options are names inside each sample directory.")
    sub.add_argument(dest="sampleDirsFile", type=str, help="Relative or absolute path to file containing a list of directories -- one per sample")
    sub.add_argument("--pileupName", dest="pileupName", type=str, default="reads.all.pileup", metavar="NAME", help="File name of the genome-wide pileup file in each sample directory.")
    sub.add_argument("-f", "--force", dest="forceFlag", action="store_true", help="Force processing even when result file already exists and is newer than inputs.")
    sub.add_argument("-l", "--snpListFile", dest="snpListFile", type=str, default="snplist.txt", metavar="FILE", help="Relative or absolute path to the SNP list file across all samples.")
    sub.add_argument("-e", "--excludeFile", dest="excludeFile", type=str, default=None, metavar="NAME", help="File name, in each sample directory, of the VCF file of positions to exclude.")
    sub.add_argument("-o", "--output", dest="consensusFile", type=str, default="consensus.fasta", metavar="NAME", help="Output file name of the consensus fasta file in each sample directory.")
    sub.add_argument("-q", "--minBaseQual", dest="minBaseQual", type=int, default=0, metavar="INT", help="Mimimum base quality score to count a read.")
    sub.add_argument("-c", "--minConsFreq", dest="minConsFreq", type=_min_cons_freq, default=0.60, metavar="FREQ", help="Consensus frequency.")
    sub.add_argument("-D", "--minConsDpth", dest="minConsDpth", type=int, default=1, metavar="INT", help="Consensus depth.")
    sub.add_argument("-d", "--minConsStrdDpth", dest="minConsStrdDpth", type=int, default=0, metavar="INT", help="Consensus strand depth.")
    sub.add_argument("-b", "--minConsStrdBias", dest="minConsStrdBias", type=_min_cons_strand_bias, default=0, metavar="FREQ", help="Strand bias.")
    sub.add_argument("--vcfFileName", dest="vcfFileName", type=str, default=None, metavar="NAME", help="VCF Output file name.")
    sub.add_argument("--vcfRefName", dest="vcfRefName", type=str, default="Unknown reference", metavar="NAME", help="Name of the reference file.  This is only used in the generated VCF file header.")
    sub.add_argument("--vcfAllPos", dest="vcfAllPos", action="store_true", help="Flag to cause VCF file generation at all positions, not just the snp positions.")
    sub.add_argument("--vcfPreserveRefCase", dest="vcfPreserveRefCase", action="store_true", help="Emit each reference base in uppercase/lowercase as it appears in the reference sequence file.")
    sub.add_argument("--vcfFailedSnpGt", dest="vcfFailedSnpGt", type=str, default=".", choices=[".", "0", "1"], help="Controls the VCF file GT data element when a snp fails filters.")
    sub.add_argument("--amdMetricsRefFasta", dest="amdMetricsRefFasta", type=str, default=None, metavar="FILE", help="Extension of this build: reference fasta file; when given, the mean pileup depth (a by-product of the pileup scan) and the number of missing positions are recorded in the sample's metrics file, where the reference's collect_metrics reuses them instead of reading the pileup again.")
    sub.add_argument("--amdMetricsFile", dest="amdMetricsFile", type=str, default=None, metavar="FILE", help="Extension of this build: the metrics file to update (default: metrics in the sample directory).")
    _common(sub)
    sub.set_defaults(func=call_consensus.call_consensus_batch, excepthook=utils.handle_global_exception)

    # Extension of this build (no reference counterpart): steps 4-11 as one job, every pileup over the host link once
    sub = subparsers.add_parser("hot_path_batch", help="site calling to distance matrices for all samples as one job, one rank per GPU", formatter_class=fmt,
                                description="Run the post-alignment steps of the pipeline (site calling, filter_regions, merge_sites, call_consensus, snp_matrix, snp_reference, distance — for the original and the preserved SNP lists) for every sample directory of sampleDirsFile as one job: each pileup is copied to the GPU once and stays there between the steps.  Start one process per GPU with torchrun for more than one GPU.  Writes the same files as the separate steps.")
    hot_path.add_arguments(sub)
    _common(sub)
    sub.set_defaults(func=hot_path.hot_path_batch, excepthook=utils.handle_global_exception)

    # Extension of this build (no reference counterpart): the per-node service behind the per-sample CLI (SNPGPU_SERVICE)
    sub = subparsers.add_parser("serve", help="keep the GPU context for the per-sample subcommands of this node", formatter_class=fmt,
                                description="Start the per-node service: one worker process per GPU that keeps the device context and the pinned staging buffers, and runs the subcommands that cfsan_snp_pipeline processes started with SNPGPU_SERVICE set pass on to it.")
    service.add_arguments(sub)
    _common(sub)
    sub.set_defaults(func=service.serve, excepthook=utils.handle_global_exception)

    sub = subparsers.add_parser("snp_matrix", help="Create a matrix of SNPs", formatter_class=fmt,
                                description="Create the SNP matrix containing the consensus base for each of the samples at the positions where high-confidence SNPs were found in any of the samples.")
    sub.add_argument(dest="sampleDirsFile", type=str, help="Relative or absolute path to file containing a list of directories -- one per sample")
    sub.add_argument("-f", "--force", dest="forceFlag", action="store_true", help="Force processing even when result file already exists and is newer than inputs")
    sub.add_argument("-c", "--consFileName", dest="consFileName", type=str, default="consensus.fasta", metavar="NAME", help="File name of the previously created consensus SNP call file which must exist in each of the sample directories")
    sub.add_argument("-o", "--output", dest="snpmaFile", type=str, default="snpma.fasta", metavar="FILE", help="Output file.  Relative or absolute path to the SNP matrix file")
    _common(sub)
    sub.set_defaults(func=snp_matrix.create_snp_matrix, excepthook=utils.handle_global_exception)

    sub = subparsers.add_parser("distance", help="Calculate the SNP distances between samples", formatter_class=fmt,
                                description="Calculate pairwise SNP distances from the multi-fasta SNP matrix. Generates a file of pairwise distances and a file containing a matrix of distances.")
    sub.add_argument(dest="inputFile", type=str, metavar="snpMatrixFile", help="Relative or absolute path to the input multi-fasta SNP matrix file.")
    sub.add_argument("-f", "--force", dest="forceFlag", action="store_true", help="Force processing even when result file already exists and is newer than inputs")
    sub.add_argument("-p", "--pairs", dest="pairwiseFile", type=str, default=None, metavar="FILE", help="Relative or absolute path to the pairwise distance output file.")
    sub.add_argument("-m", "--matrix", dest="matrixFile", type=str, default=None, metavar="FILE", help="Relative or absolute path to the distance matrix output file.")
    _common(sub)
    sub.set_defaults(func=distance.calculate_snp_distances, excepthook=utils.handle_global_exception)

    sub = subparsers.add_parser("snp_reference", help="Write reference bases at SNP locations to a fasta file", formatter_class=fmt,
                                description="Write reference sequence bases at SNP locations to a fasta file.")
    sub.add_argument(dest="referenceFile", type=str, help="Relative or absolute path to the reference bases file in fasta format")
    sub.add_argument("-f", "--force", dest="forceFlag", action="store_true", help="Force processing even when result file already exists and is newer than inputs")
    sub.add_argument("-l", "--snpListFile", dest="snpListFile", type=str, default="snplist.txt", metavar="FILE", help="Relative or absolute path to the SNP list file")
    sub.add_argument("-o", "--output", dest="snpRefFile", type=str, default="referenceSNP.fasta", metavar="FILE", help="Output file.  Relative or absolute path to the SNP reference sequence file")
    _common(sub)
    sub.set_defaults(func=snp_reference.create_snp_reference_seq, excepthook=utils.handle_global_exception)

    for name in NOT_PROVIDED:
        sub = subparsers.add_parser(name, help="(not part of this build)", add_help=False)
        sub.add_argument("rest", nargs=argparse.REMAINDER)
        sub.add_argument("-v", "--verbose", dest="verbose", type=int, default=1)
        sub.set_defaults(func=_not_provided, excepthook=utils.handle_global_exception)

    args = parser.parse_args(argv)

    if args.subparser_name == "filter_regions":              # cfsan_snp_pipeline.py:530-543
        if len(args.windowSizeList) != len(args.maxSnpsList):
            utils.global_error("Error: you must specify the same number of arguments for window size and max snps.")
        for window_size in args.windowSizeList:
            if window_size < 1:
                utils.global_error("Error: the length of the window must be a positive integer, and the input is %d." % window_size)
        for max_snps in args.maxSnpsList:
            if max_snps < 1:
                utils.global_error("Error: the maximum number of SNPs allowed must be a positive integer, and the input is %d." % max_snps)
        if args.edgeLength < 1:
            utils.global_error("Error: the length of the edge regions must be a positive integer, and the input is %d." % args.edgeLength)
    return args


def parse_command_line(line):
    return parse_argument_list(line.split())


def run_command_from_args(args):
    """Run a subcommand with previously parsed arguments.  Returns 0 on success."""
    if args.excepthook:
        sys.excepthook = args.excepthook
    utils.set_logging_verbosity(args)
    args.func(args)
    from . import timing
    timing.mark("done")
    timing.report(args.subparser_name)
    verbose_print("")
    verbose_print("# %s %s %s finished" % (utils.timestamp(), utils.program_name(), args.subparser_name))
    return 0


def run_command_from_arg_list(argv):
    return run_command_from_args(parse_argument_list(argv))


def run_command_from_line(line):
    return run_command_from_arg_list(line.split())


def main():
    """Console entry point.  A CLI process never exchanges device pointers with torch, so the HIP library is loaded
    without importing it (seconds per sample process); see _lib.load()."""
    if len(sys.argv) > 1 and sys.argv[1] in NOT_PROVIDED:    # with bin/ in front of PATH, `cfsan_snp_pipeline run ...` still works
        other = reference_cli()
        if other:
            sys.stdout.flush()
            os.execv(other, [other] + sys.argv[1:])
    from . import _lib
    _lib.TORCH_FREE_OK = True
    return run_command_from_arg_list(sys.argv[1:])


if __name__ == "__main__":
    sys.exit(main())

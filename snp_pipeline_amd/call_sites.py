"""call_sites subcommand: reads.all.pileup and var.flt.vcf for one sample.

Host mirror of snppipeline/call_sites.py:15-111.  The pileup is still made by ``samtools mpileup`` (an external tool on both
sides, untouched); the site calling that the reference hands to ``java -jar VarScan.jar mpileup2snp`` (call_sites.py:89-108)
runs on the device instead (varscan.py / csrc/varscan.hip), honouring the same ``VarscanMpileup2snp_ExtraParams``.
"""
from __future__ import print_function

import os
import subprocess

from . import utils
from . import varscan
from .utils import verbose_print


def _add_file_suffix(path, suffix, enable=True):
    """utils.add_file_suffix (utils.py:286-330): the suffix goes before the extension."""
    if not enable:
        return path
    root, ext = os.path.splitext(path)
    return root + suffix + ext


def _sample_error_on_missing_file(file_path, program, empty_ok=False):
    """utils.sample_error_on_missing_file (utils.py:930-951)."""
    if not os.path.isfile(file_path):
        utils.sample_error("Error: %s does not exist after running %s." % (file_path, program))
    if not empty_ok and os.path.getsize(file_path) == 0:
        utils.sample_error("Error: %s is empty after running %s." % (file_path, program))


def call_sites(args):
    """Entry point of ``cfsan_snp_pipeline call_sites`` (cfsan_snp_pipeline.py:294-304)."""
    utils.print_log_header(classpath=True)
    utils.print_arguments(args)

    reference_file_path = args.referenceFile
    utils.verify_non_empty_input_files("Reference file", [reference_file_path], error_handler="global")
    sample_dir = args.sampleDir

    remove_duplicate_reads = os.environ.get("RemoveDuplicateReads", "true").lower() == "true"
    enable_local_realignment = os.environ.get("EnableLocalRealignment", "true").lower() == "true"
    input_bam_file = os.path.join(sample_dir, "reads.sorted.bam")
    input_bam_file = _add_file_suffix(input_bam_file, ".deduped", enable=remove_duplicate_reads)
    input_bam_file = _add_file_suffix(input_bam_file, ".indelrealigned", enable=enable_local_realignment)
    utils.verify_non_empty_input_files("Sample BAM file", [input_bam_file], error_handler="sample")
    sample_id = os.path.basename(os.path.abspath(sample_dir))

    # ---- the pileup: samtools, exactly as the reference runs it (call_sites.py:68-83) ----
    pileup_file = os.path.join(sample_dir, "reads.all.pileup")
    needs_rebuild = utils.target_needs_rebuild([input_bam_file, reference_file_path], pileup_file)
    if not args.forceFlag and not needs_rebuild:
        verbose_print("# Pileup file is already freshly created for %s.  Use the -f option to force a rebuild." % sample_id)
    else:
        extra = os.environ.get("SamtoolsMpileup_ExtraParams") or ""
        command_line = "samtools mpileup " + extra + " -f " + reference_file_path + " " + input_bam_file
        verbose_print("# Create pileup from bam file.")
        verbose_print("# %s %s" % (utils.timestamp(), command_line))
        with open(pileup_file, "w") as out:
            subprocess.check_call(command_line, shell=True, stdout=out)
        _sample_error_on_missing_file(pileup_file, "samtools mpileup")
        verbose_print("")

    # ---- the sites: mpileup2snp on the device (call_sites.py:89-108) ----
    vcf_file = os.path.join(sample_dir, "var.flt.vcf")
    needs_rebuild = utils.target_needs_rebuild([pileup_file], vcf_file)
    if not args.forceFlag and not needs_rebuild:
        verbose_print("# VCF file is already freshly created for %s.  Use the -f option to force a rebuild." % sample_id)
    else:
        extra = os.environ.get("VarscanMpileup2snp_ExtraParams") or ""
        opts = varscan.Options(extra)
        verbose_print("# Create vcf file")
        verbose_print("# %s mpileup2snp (device) %s --output-vcf 1 %s" % (utils.timestamp(), pileup_file, extra))
        from .device import default_device
        n_lines, n_rows = varscan.mpileup2snp(default_device(), pileup_file, vcf_file, opts)
        verbose_print("# %d pileup lines, %d variant sites" % (n_lines, n_rows))
        _sample_error_on_missing_file(vcf_file, "VarScan")

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


def _bam_and_pileup(sample_dir):
    """The BAM the pileup is made from (its name follows RemoveDuplicateReads / EnableLocalRealignment, call_sites.py:55-59)
    and the pileup's path."""
    remove_duplicate_reads = os.environ.get("RemoveDuplicateReads", "true").lower() == "true"
    enable_local_realignment = os.environ.get("EnableLocalRealignment", "true").lower() == "true"
    input_bam_file = os.path.join(sample_dir, "reads.sorted.bam")
    input_bam_file = _add_file_suffix(input_bam_file, ".deduped", enable=remove_duplicate_reads)
    input_bam_file = _add_file_suffix(input_bam_file, ".indelrealigned", enable=enable_local_realignment)
    return input_bam_file, os.path.join(sample_dir, "reads.all.pileup")


def call_sites(args):
    """Entry point of ``cfsan_snp_pipeline call_sites`` (cfsan_snp_pipeline.py:294-304)."""
    utils.print_log_header(classpath=True)
    utils.print_arguments(args)

    reference_file_path = args.referenceFile
    utils.verify_non_empty_input_files("Reference file", [reference_file_path], error_handler="global")
    sample_dir = args.sampleDir

    input_bam_file, pileup_file = _bam_and_pileup(sample_dir)
    utils.verify_non_empty_input_files("Sample BAM file", [input_bam_file], error_handler="sample")
    sample_id = os.path.basename(os.path.abspath(sample_dir))

    # ---- the pileup: samtools, exactly as the reference runs it (call_sites.py:68-83) ----
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
        open(vcf_file, "w").close()                          # as command.run does for its target: a failed pass leaves no stale var.flt.vcf behind
        n_lines, n_rows = varscan.mpileup2snp(default_device(), pileup_file, vcf_file, opts)
        verbose_print("# %d pileup lines, %d variant sites" % (n_lines, n_rows))
        _sample_error_on_missing_file(vcf_file, "VarScan")


def call_sites_batch(args):
    """``cfsan_snp_pipeline call_sites_batch`` — an extension of this build, not a reference subcommand: the call_sites step
    (run.py:672-702 starts one process per sample) for every sample directory of sampleDirsFile in one process.  Stale
    pileups are made with ``samtools mpileup`` exactly as call_sites does (a few at a time); then the samples are dealt
    round-robin to the visible GPUs, one host thread per GPU, each with ONE streamed device call for all its pileups
    (snpgpu_varscan_files).  A failing sample is reported as a sample error and the others still run."""
    import concurrent.futures
    import threading
    from . import device as devmod
    utils.print_log_header(classpath=True)
    utils.print_arguments(args)
    reference_file_path = args.referenceFile
    utils.verify_non_empty_input_files("Reference file", [reference_file_path], error_handler="global")
    if utils.verify_non_empty_input_files("File of sample directories", [args.sampleDirsFile]) > 0:
        utils.global_error(None)
    with open(args.sampleDirsFile, "r") as f:
        sample_dirs = [d for d in (line.rstrip() for line in f) if d]
    extra_samtools = os.environ.get("SamtoolsMpileup_ExtraParams") or ""
    opts = varscan.Options(os.environ.get("VarscanMpileup2snp_ExtraParams") or "")

    failed = 0
    stale, todo = [], []
    for sample_dir in sample_dirs:
        bam, pileup_file = _bam_and_pileup(sample_dir)
        if utils.verify_non_empty_input_files("Sample BAM file", [bam]) > 0:
            utils.sample_error("Error: cannot call sites without the sample BAM file.", continue_possible=True)
            failed += 1
            continue
        if args.forceFlag or utils.target_needs_rebuild([bam, reference_file_path], pileup_file):
            stale.append((sample_dir, bam, pileup_file))
        todo.append((sample_dir, pileup_file, os.path.join(sample_dir, "var.flt.vcf")))

    def make_pileup(item):
        _, bam, pileup_file = item
        command_line = "samtools mpileup " + extra_samtools + " -f " + reference_file_path + " " + bam
        with open(pileup_file, "wb") as out:
            return subprocess.call(command_line, shell=True, stdout=out)

    bad = set()
    if stale:
        verbose_print("# %s samtools mpileup for %d samples" % (utils.timestamp(), len(stale)))
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            for (sample_dir, _, pileup_file), rc in zip(stale, ex.map(make_pileup, stale)):
                if rc != 0 or not os.path.isfile(pileup_file) or os.path.getsize(pileup_file) == 0:
                    utils.sample_error("Error: %s is missing or empty after running samtools mpileup." % pileup_file, continue_possible=True)
                    bad.add(sample_dir)
                    failed += 1
    todo = [t for t in todo if t[0] not in bad and (args.forceFlag or utils.target_needs_rebuild([t[1]], t[2]))]
    if todo:
        pinned = os.environ.get("SNPGPU_DEVICE", os.environ.get("LOCAL_RANK"))
        devices = [int(pinned)] if pinned is not None else list(range(max(1, devmod.device_count())))
        devices = devices[:len(todo)]
        errors, lock = [], threading.Lock()

        def worker(dev_index, mine):
            dev = None
            try:
                dev = devmod.Device(dev_index)
                res = varscan.mpileup2snp_files(dev, [t[1] for t in mine], [t[2] for t in mine], opts)
                with lock:
                    for t, r in zip(mine, res):
                        if isinstance(r, Exception):
                            errors.append((t, r))
                        else:
                            verbose_print("# %s: %d pileup lines, %d variant sites" % (t[0], r[0], r[1]))
            except Exception as err:                            # noqa: B902 — the whole device stream failed: every sample of it is reported
                with lock:
                    errors.extend((t, err) for t in mine)
            finally:
                if dev is not None:
                    dev.close()

        threads = [threading.Thread(target=worker, args=(dv, todo[i::len(devices)])) for i, dv in enumerate(devices)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for t, err in errors:
            utils.sample_error("Error: call_sites failed for sample %s: %s: %s" % (os.path.basename(os.path.abspath(t[0])), type(err).__name__, err),
                               continue_possible=True)
        failed += len(errors)
    if failed:
        verbose_print("%d of %d samples failed." % (failed, len(sample_dirs)))

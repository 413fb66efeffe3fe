"""call_sites subcommand: reads.all.pileup and var.flt.vcf for one sample.

Host side of snppipeline/call_sites.py:15-111; the prologue of ``call_sites`` (input checks, the BAM's name, the samtools command
line, the freshness tests and every message) is the CLI contract and follows call_sites.py:40-92 statement by statement.  The pileup is still made by ``samtools mpileup`` (an external tool on both
sides, untouched).  The site calling has two routes, chosen by ``SNPGPU_SITE_CALLING`` (or ``--siteCalling`` of the batch
subcommands):

    varscan    ``java -jar VarScan.jar mpileup2snp <pileup> --output-vcf 1 ...`` with the jar found on CLASSPATH, exactly as
               the reference runs it (call_sites.py:89-108) — the route whose output every downstream step of this build is
               pinned against; no device is touched
    device     the restatement of mpileup2snp on the device (varscan.py / csrc/varscan.hip), honouring the same
               ``VarscanMpileup2snp_ExtraParams``; its read counting is PARITY UNPINNED (no VarScan and no (pileup, var.flt.vcf)
               pair exists in the reference tree to pin it: DESIGN.md section 2)
    existing   var.flt.vcf is an INPUT: it is never written; a sample without one (or with one older than its pileup) is a
               sample error.  For trees whose site calling was done elsewhere — by the reference, by real VarScan
    auto       (the default) ``varscan`` when a VarScan jar is on CLASSPATH — what a working installation of the reference
               has, so dropping this build in changes nothing upstream of the pinned steps — else ``device`` (where the
               reference would stop with "cannot execute VarScan")
"""
from __future__ import print_function

import os
import subprocess
import sys

from . import utils
from . import varscan
from .utils import verbose_print

SITE_CALLING_MODES = ("auto", "varscan", "device", "existing")


def find_path_in_path_list(search_item, env_var, case_sensitive=False):
    """utils.find_path_in_path_list (utils.py:1512-1563): the first entry of a colon-separated variable that contains the item."""
    if not case_sensitive:
        search_item = search_item.lower()
    for path in os.environ.get(env_var, "").split(":"):
        if search_item in (path if case_sensitive else path.lower()):
            return path
    return None


def site_calling_mode(given=None):
    """The route for this process: `given` (a --siteCalling option) or $SNPGPU_SITE_CALLING or auto; auto resolved here."""
    mode = (given or os.environ.get("SNPGPU_SITE_CALLING") or "auto").lower()
    if mode not in SITE_CALLING_MODES:
        utils.global_error("Error: site calling mode must be one of %s, not %r." % (", ".join(SITE_CALLING_MODES), mode))
    if mode == "auto":
        mode = "varscan" if find_path_in_path_list("VarScan", "CLASSPATH") else "device"
    return mode


DEVICE_PASS_NOTE = ("# site calling: device pass (parity unpinned) - no VarScan jar on CLASSPATH; the read counting and line selection of "
                    "csrc/varscan.hip follow oracle/varscan_oracle.py, which no file of the reference pins.  Put the jar on CLASSPATH "
                    "(mode varscan) or supply var.flt.vcf (mode existing) for the reference's own sites.")


def log_site_calling_mode(mode):
    """One line in the step log naming the route; whenever it is the device pass, say that its parity is unpinned."""
    verbose_print("# site calling mode: %s" % mode)
    if mode == "device":
        verbose_print(DEVICE_PASS_NOTE)


def _run(command_line, outfile=None):
    """command.run (command.py:17-88): a shell command with stdout captured or written to a file, stderr inherited;
    CalledProcessError for a non-zero exit when writing to a file."""
    sys.stdout.flush()
    if outfile is None:
        proc = subprocess.Popen(command_line, stdout=subprocess.PIPE, shell=True)
        out, _ = proc.communicate()
        return out.decode(sys.stdout.encoding or sys.stdin.encoding or "utf-8")
    with open(outfile, "wb") as out:
        subprocess.check_call(command_line, stdout=out, shell=True)
    return None


def extract_version_str(program_name, command_line):
    """utils.extract_version_str (utils.py:188-228)."""
    lines = [ln for ln in (ln.strip() for ln in _run(command_line).split("\n")) if ln]
    for line in lines:
        lower = line.lower()
        if "version" in lower:
            tokens = lower.replace(":", " ").split()
            for index, token in enumerate(tokens):
                if token == "version" and len(tokens) > index + 1:
                    return program_name + " version " + tokens[index + 1]
    if len(lines) == 1 and len(lines[0].split()) == 1:
        return program_name + " version " + lines[0].split()[0]
    return "Unrecognized " + program_name + " version"


def varscan_command_line(jar_file_path, pileup_file):
    """The command of call_sites.py:96-98, character for character."""
    jvm_extra = os.environ.get("VarscanJvm_ExtraParams") or ""
    extra = os.environ.get("VarscanMpileup2snp_ExtraParams") or ""
    return "java " + jvm_extra + " -jar " + jar_file_path + " mpileup2snp " + pileup_file + " --output-vcf 1 " + extra


def _file_contains(file_path, text):
    with open(file_path, errors="replace") as f:
        return any(text in line for line in f)


def run_varscan_jar(pileup_file, vcf_file, error=None, log=True):
    """call_sites.py:89-108: find the jar, run mpileup2snp into vcf_file, then the three checks of the result.
    error(message): how a failed check is reported (default utils.sample_error, which ends the process as the reference's does);
    the batch callers collect the message instead."""
    error = error or utils.sample_error
    jar_file_path = find_path_in_path_list("VarScan", "CLASSPATH")
    if not jar_file_path:
        utils.global_error("Error: cannot execute VarScan. Define the path to VarScan.jar in the CLASSPATH environment variable.")
    command_line = varscan_command_line(jar_file_path, pileup_file)
    if log:
        version_str = extract_version_str("VarScan", "java -jar " + jar_file_path + " 2>&1 > /dev/null | head -n 1 | cut -d ' ' -f 2")
        verbose_print("# Create vcf file")
        verbose_print("# %s %s" % (utils.timestamp(), command_line))
        verbose_print("# %s" % version_str)
    _run(command_line, vcf_file)
    if not os.path.isfile(vcf_file):
        return error("Error: %s does not exist after running %s." % (vcf_file, "VarScan"))
    if os.path.getsize(vcf_file) == 0:
        return error("Error: %s is empty after running %s." % (vcf_file, "VarScan"))
    for text in ("OutOfMemoryError", "Insufficient"):         # utils.sample_error_on_file_contains (utils.py:954-974)
        if _file_contains(vcf_file, text):
            return error("Error: %s contains unexpected text: '%s' after running %s." % (vcf_file, text, "VarScan"))
    return None


def run_varscan_jar_many(items, max_workers=None):
    """The jar for many samples at once — items: [(pileup_file, vcf_file)] — on a few host threads (each is a JVM process; the
    reference's job array runs ``max_cpu_cores`` of them at a time, run.py:662-664).  Returns one entry per item: None, or the
    message of that sample's failure (a failed check of the result, a non-zero exit of java)."""
    import concurrent.futures
    if not find_path_in_path_list("VarScan", "CLASSPATH"):
        utils.global_error("Error: cannot execute VarScan. Define the path to VarScan.jar in the CLASSPATH environment variable.")

    def one(item):
        pileup_file, vcf_file = item
        try:
            return run_varscan_jar(pileup_file, vcf_file, error=lambda message: message, log=False)
        except subprocess.CalledProcessError as err:
            return "Error: %s returned non-zero exit status %d." % (err.cmd, err.returncode)
        except (IOError, OSError) as err:
            return "Error: cannot run VarScan for %s: %s" % (pileup_file, err)

    if not items:
        return []
    from . import device as devmod
    workers = max_workers or devmod.host_threads(len(items), share=2)     # (a JVM per worker: half of this process's CPU budget)
    with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as ex:
        return list(ex.map(one, items))


def check_existing_vcf(pileup_file, vcf_file):
    """Site calling mode 'existing': the message for a var.flt.vcf that cannot serve as input, or None."""
    if not os.path.isfile(vcf_file):
        return "Error: VCF file %s does not exist (site calling mode 'existing' never writes it)." % vcf_file
    if os.path.getsize(vcf_file) == 0:
        return "Error: VCF file %s is empty (site calling mode 'existing' never writes it)." % vcf_file
    if utils.target_needs_rebuild([pileup_file], vcf_file):
        return "Error: %s is older than %s (site calling mode 'existing' never rebuilds it)." % (vcf_file, pileup_file)
    return None


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
        version_str = extract_version_str("SAMtools", "samtools 2>&1 > /dev/null")
        verbose_print("# Create pileup from bam file.")
        verbose_print("# %s %s" % (utils.timestamp(), command_line))
        verbose_print("# %s" % version_str)
        _run(command_line, pileup_file)
        _sample_error_on_missing_file(pileup_file, "samtools mpileup")
        verbose_print("")

    # ---- the sites (call_sites.py:85-108): the VarScan jar as the reference runs it, or the device pass ----
    vcf_file = os.path.join(sample_dir, "var.flt.vcf")
    mode = site_calling_mode()
    if mode == "existing":
        if utils.verify_non_empty_input_files("VCF file", [vcf_file]) > 0:
            utils.sample_error("Error: site calling mode 'existing' needs the var.flt.vcf of sample %s." % sample_id)
        if utils.target_needs_rebuild([pileup_file], vcf_file):
            utils.sample_error("Error: %s is older than %s (site calling mode 'existing' never rebuilds it)." % (vcf_file, pileup_file))
        verbose_print("# VCF file of %s is an input (site calling mode 'existing')." % sample_id)
        return
    needs_rebuild = utils.target_needs_rebuild([pileup_file], vcf_file)
    if not args.forceFlag and not needs_rebuild:
        verbose_print("# VCF file is already freshly created for %s.  Use the -f option to force a rebuild." % sample_id)
    elif mode == "varscan":
        run_varscan_jar(pileup_file, vcf_file)
    else:
        extra = os.environ.get("VarscanMpileup2snp_ExtraParams") or ""
        opts = varscan.Options(extra)
        verbose_print("# Create vcf file")
        verbose_print(DEVICE_PASS_NOTE)
        verbose_print("# %s mpileup2snp (device) %s --output-vcf 1 %s" % (utils.timestamp(), pileup_file, extra))
        from .device import default_device
        open(vcf_file, "w").close()                          # as command.run does for its target: a failed pass leaves no stale var.flt.vcf behind
        n_lines, n_rows = varscan.mpileup2snp(default_device(), pileup_file, vcf_file, opts)
        verbose_print("# %d pileup lines, %d variant sites" % (n_lines, n_rows))
        _sample_error_on_missing_file(vcf_file, "VarScan")


def call_sites_batch(args):
    """``cfsan_snp_pipeline call_sites_batch`` — an extension of this build, not a reference subcommand: the call_sites step
    (run.py:672-702 starts one process per sample) for every sample directory of sampleDirsFile in one process.  Stale
    pileups are made with ``samtools mpileup`` exactly as call_sites does (a few at a time); then, by site calling mode (the
    module docstring; ``--siteCalling``): ``device`` — the samples are dealt round-robin to the visible GPUs, one host thread per
    GPU, each with ONE streamed device call for all its pileups (snpgpu_varscan_files); ``varscan`` — the VarScan jar per stale
    sample, a few JVMs at a time; ``existing`` — the var.flt.vcf files are only checked.  A failing sample is reported as a
    sample error and the others still run."""
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
        from . import device as devmod
        with concurrent.futures.ThreadPoolExecutor(max_workers=devmod.host_threads(8)) as ex:
            for (sample_dir, _, pileup_file), rc in zip(stale, ex.map(make_pileup, stale)):
                if rc != 0 or not os.path.isfile(pileup_file) or os.path.getsize(pileup_file) == 0:
                    utils.sample_error("Error: %s is missing or empty after running samtools mpileup." % pileup_file, continue_possible=True)
                    bad.add(sample_dir)
                    failed += 1
    mode = site_calling_mode(getattr(args, "siteCalling", None))
    log_site_calling_mode(mode)
    todo = [t for t in todo if t[0] not in bad]
    if mode == "existing":                                       # var.flt.vcf is an input: checked, never written
        for t in todo:
            message = check_existing_vcf(t[1], t[2])
            if message:
                utils.sample_error(message, continue_possible=True)
                failed += 1
        todo = []
    todo = [t for t in todo if args.forceFlag or utils.target_needs_rebuild([t[1]], t[2])]
    if todo and mode == "varscan":                               # the jar, as call_sites.py:89-108 runs it, a few JVMs at a time
        jar = find_path_in_path_list("VarScan", "CLASSPATH")
        verbose_print("# %s %s  (and so on: %d samples)" % (utils.timestamp(), varscan_command_line(jar or "VarScan.jar", todo[0][1]), len(todo)))
        for t, message in zip(todo, run_varscan_jar_many([(t[1], t[2]) for t in todo])):
            if message:
                utils.sample_error(message, continue_possible=True)
                failed += 1
        todo = []
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

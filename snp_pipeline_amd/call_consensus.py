"""call_consensus subcommand: consensus base of one sample at every snplist position.

Host mirror of snppipeline/call_consensus.py:18-192.  Argument handling, input checks, freshness and the FASTA/VCF
writers live here; reading the pileup, the per-position base counting and the caller's filters run in the HIP
kernels (csrc/scan.hip, csrc/consensus.hip) behind ``Device.call_consensus_files``: the pileup FILE is streamed to the
device in chunks (csrc/stream.hip) and scanned while it arrives; it is never resident in host memory.

``call_consensus_batch`` is the same step for every sample of a sampleDirsFile in ONE process: what run.py:704-718
starts as an array of per-sample processes becomes one stream of files per visible GPU (samples dealt round-robin).
"""
from __future__ import print_function

import os
import threading

import numpy as np

from . import _lib as L
from . import device as devmod
from . import timing
from . import utils
from . import vcf_writer


def _unique_sites(arrays):
    """The distinct (contig, position) pairs of an array triple (the size of the reference's set of them)."""
    _, cidx, pos = arrays
    return np.unique((cidx.astype(np.int64) << 40) ^ pos) if len(pos) else np.zeros(0, np.int64)


def build_siteset(dev, snp_arrays, excluded_arrays=None):
    """snp_arrays / excluded_arrays: (contig names, contig index per record, position per record) as
    utils.read_snp_position_arrays / read_vcf_site_arrays return them (snplist order / the exclude VCF's records).  Returns
    (SiteSet over their union — call_consensus.py:147-151 —, the slot of every snplist line, the slots of the exclude list).
    numpy all the way: no Python loop per key."""
    lists = [snp_arrays + (L.SITE_IN_SNPLIST,)]
    if excluded_arrays is not None:
        lists.append(excluded_arrays + (L.SITE_EXCLUDED,))
    ss, slots = dev.siteset_from_lists(lists)
    return ss, slots[0], (slots[1] if excluded_arrays is not None else None)


def tuples_to_arrays(sites):
    """[(chrom str or bytes, pos)] -> the array triple of utils.read_vcf_site_arrays (tests and host-buffer callers)."""
    return utils._site_arrays_from_tuples(sites)


def consensus_string(snp_slots, res):
    """The consensus bytes in snplist order: '-' for a position that cannot occur in a pileup (no slot)."""
    return np.where(snp_slots >= 0, res.bases[np.maximum(snp_slots, 0)], 0x2D).astype(np.uint8)


def consensus_for_sample(dev, pileup_bytes, snp_list, excluded_positions, params, want_counts=False):
    """Host-buffer form (tests, bench): returns (consensus str in snplist order, SiteSet, ConsensusResult)."""
    ss, snp_slots, _ = build_siteset(dev, tuples_to_arrays(snp_list), tuples_to_arrays(sorted(excluded_positions)))
    res = dev.call_consensus(ss, pileup_bytes, params, want_counts=want_counts)
    return consensus_string(snp_slots, res).tobytes().decode("ascii"), ss, res


class _ValidUtf8(devmod.PileupFormatError):
    """The pileup holds bytes >= 0x80 and is valid UTF-8."""

    def __init__(self, message):
        devmod.PileupFormatError.__init__(self, message, None)


def _raise_as_reference(err, pileup_path=None, every_line_is_a_record=False):
    """Re-raise a device-detected malformed pileup as the exception class the reference raises for it, so that the
    error log names the same exception type (utils.handle_sample_exception prints ``exc_type.__name__``).
    every_line_is_a_record: the --vcfAllPos reader (pileup.py:418-421) builds a Record from every line, and a line with fewer
    than two fields then ends with IndexError (pileup.py:223-224) where the reader with a position set fails to unpack two
    values (ValueError, pileup.py:425)."""
    exc = getattr(err, "reference_exception", None)
    if every_line_is_a_record and getattr(err, "scan_code", 0) == 1:
        exc = IndexError
    if getattr(err, "scan_code", 0) == 3 and pileup_path:
        # A byte >= 0x80.  The reference reads the pileup as text (pileup.py:405, the locale's encoding: UTF-8 on the
        # pipeline's platforms), so a file that is not valid UTF-8 ends its run with UnicodeDecodeError: the same here.
        # (Valid multi-byte characters are the one input that is still refused: they would count as single symbols.)
        with open(pileup_path, "rb") as f:
            f.read().decode("utf-8")
        raise _ValidUtf8(str(err))                   # (call_consensus tries the escaped-names bridge, utf8_names.py, before giving up)
    if exc is None:
        raise err
    raise exc(str(err))


class _Plan(object):
    """Everything call_consensus.py:100-140 derives from the arguments of one sample."""

    def __init__(self, args, all_pileup_file_path, consensus_file_path, exclude_file_path):
        self.args = args
        self.pileup_path = all_pileup_file_path
        self.read_path = all_pileup_file_path         # what the device reads: the pileup, or its copy with escaped contig names (utf8_names.py)
        self.consensus_path = consensus_file_path
        self.exclude_path = exclude_file_path
        self.sample_name = os.path.basename(os.path.dirname(os.path.abspath(all_pileup_file_path)))
        consensus_file_dir = os.path.dirname(os.path.abspath(consensus_file_path))
        self.vcf_path = os.path.join(consensus_file_dir, args.vcfFileName) if args.vcfFileName else None
        self.excluded = None                 # the exclude VCF's (contig names, contig index, position) arrays
        self.batch = False


def _write_outputs(plan, dev, ss, snp_slots, res, file_flags=None):
    """file_flags: the site flags this file was called with (the set's, plus its own exclude list in a batch); rows of the
    VCF are the parsed positions of THIS sample (a batch's set also holds the other samples' exclude positions)."""
    args = plan.args
    if file_flags is None:
        file_flags = ss.flags
    in_snplist = (file_flags & L.SITE_IN_SNPLIST) != 0
    utils.verbose_print("called consensus positions = %i" % int(((res.counts["status"] != L.ST_NO_LINE) & in_snplist).sum()))
    if plan.vcf_path:
        dup = res.n_matched > int(np.count_nonzero(res.line_offsets))
        if args.vcfAllPos or dup:
            # every LINE gets a record (--vcfAllPos, pileup.py:418-421), or the pileup repeats a position — the reference writes
            # a row for every matching line (call_consensus.py:178-180) where the per-site result only knows the last one: the
            # rows come from the all-lines pass, run with this sample's own flags (in a batch the shared set knows nothing of
            # its exclude list)
            params = devmod.make_params(args.minBaseQual, args.minConsFreq, args.minConsDpth, args.minConsStrdDpth, args.minConsStrdBias)
            own = ss if file_flags is ss.flags else devmod.SiteSet.from_arrays(dev, ss.contigs, ss.keys[file_flags != 0], file_flags[file_flags != 0])
            try:
                vcf_writer.write_all_positions_vcf_from_pileup(dev, own, plan.vcf_path, plan.sample_name, args, plan.read_path, params,
                                                               only_listed=not args.vcfAllPos, check=bool(args.vcfAllPos))
            except devmod.PileupFormatError as err:
                _raise_as_reference(err, plan.read_path, bool(args.vcfAllPos))
            finally:
                if own is not ss:
                    own.close()
        else:
            vcf_writer.write_consensus_vcf(plan.vcf_path, plan.sample_name, args, ss, res, res.line_offsets, parsed=file_flags != 0)
    consensus = consensus_string(snp_slots, res).tobytes().decode("ascii")
    with open(plan.consensus_path, "w") as fasta_file_object:
        utils.write_fasta_record(fasta_file_object, plan.sample_name, consensus)
    if getattr(args, "amdMetricsRefFasta", None):
        _record_metrics(plan, res, consensus)


def _record_metrics(plan, res, consensus):
    """collect_metrics by-products: the reference re-reads the whole pileup in Python to sum its depth column
    (collect_metrics.py:325-340) and re-reads the consensus FASTA to count its gaps (:109-128) — both fall out of this step.
    They are recorded in the sample's metrics file under the names collect_metrics uses; when that file is newer than the
    pileup / the FASTA, the reference's collect_metrics takes them from there (:318-321, :441-447) and skips the re-reads.
    The preserved flow (an exclude file was given) records missingPosPreserved.  An EXISTING metrics file keeps its
    modification time (collect_metrics judges every metric by the file's age: a touched file would make its other, possibly
    stale values look fresh), so the by-products only save the re-reads where this step creates the file or the file was
    fresh already; the update itself is serialised by a lock file (the two flows of a sample may run at the same time)."""
    args = plan.args
    sample_dir = os.path.dirname(os.path.abspath(plan.pileup_path))
    path = args.amdMetricsFile or os.path.join(sample_dir, "metrics")
    if plan.batch and args.amdMetricsFile:                  # in a batch the option is a NAME inside each sample directory, like -o / -e
        path = os.path.join(sample_dir, os.path.basename(args.amdMetricsFile))
    updates = {("missingPosPreserved" if plan.exclude_path else "missingPos"): str(consensus.count("-"))}
    reference_length = sum(utils.read_fasta_lengths(args.amdMetricsRefFasta).values())
    if res.depth_sum > 0 and reference_length > 0:
        updates["avePileupDepth"] = "%.2f" % (float(res.depth_sum) / float(reference_length))
    # the file keeps its modification time: collect_metrics (collect_metrics.py:318-321, :441-447) would otherwise take every
    # OTHER value already in it for fresh, also after the BAM or the VCF was rebuilt
    utils.update_properties(path, updates, keep_mtime=True)


def call_consensus(args):
    """Entry point of ``cfsan_snp_pipeline call_consensus`` (cfsan_snp_pipeline.py:345-410)."""
    utils.print_log_header()
    utils.print_arguments(args)

    snp_list_file_path = args.snpListFile
    all_pileup_file_path = args.allPileupFile
    plan = _Plan(args, all_pileup_file_path, args.consensusFile, args.excludeFile)

    if utils.verify_existing_input_files("Snplist file", [snp_list_file_path]) > 0:
        utils.global_error("Error: cannot call consensus without the snplist file.")
    if utils.verify_non_empty_input_files("Pileup file", [all_pileup_file_path]) > 0:
        utils.sample_error("Error: cannot call consensus without the pileup file.", continue_possible=False)
    source_files = [snp_list_file_path, all_pileup_file_path]

    if plan.exclude_path:
        if utils.verify_existing_input_files("Exclude file", [plan.exclude_path]) > 0:
            utils.sample_error("Error: cannot call consensus without the file of excluded positions.", continue_possible=False)
        plan.excluded = utils.read_vcf_site_arrays(plan.exclude_path)
        source_files.append(plan.exclude_path)

    if not args.forceFlag and not utils.target_needs_rebuild(source_files, plan.consensus_path):
        utils.verbose_print("Consensus call file %s has already been freshly built.  Use the -f option to force a rebuild." % plan.consensus_path)
        return

    snp_arrays = utils.read_snp_position_arrays(snp_list_file_path)
    n_excluded = len(_unique_sites(plan.excluded)) if plan.excluded is not None else 0
    utils.verbose_print("snp position list length = %d" % len(snp_arrays[2]))
    utils.verbose_print("excluded snps list length = %d" % n_excluded)
    utils.verbose_print("total snp position list length = %d" % (len(snp_arrays[2]) + n_excluded))
    timing.mark("inputs read")

    params = devmod.make_params(args.minBaseQual, args.minConsFreq, args.minConsDpth, args.minConsStrdDpth, args.minConsStrdBias)
    dev = devmod.default_device()
    timing.mark("device context")
    try:
        _call_one(plan, dev, params, snp_arrays)
    except _ValidUtf8 as err:
        _call_one_with_escaped_names(plan, dev, params, snp_arrays, err)
    timing.mark("outputs written")


def _call_one_with_escaped_names(plan, dev, params, snp_arrays, err):
    """Non-ASCII characters in a valid UTF-8 pileup.  In contig names they are just names to the reference: the device gets a copy
    of the file in which every name is escaped to ASCII (order and equality kept), the site lists likewise, and the CHROM column of
    consensus.vcf is spelled back.  Anywhere else they stay refused (utf8_names.py); a file that is not valid UTF-8 ends with
    UnicodeDecodeError, as the reference's text-mode read does."""
    from . import utf8_names
    try:
        plan.read_path = utf8_names.escaped_copy(plan.pileup_path)
    except utf8_names.Refused as why:
        raise devmod.PileupFormatError("%s (%s)" % (err, why), None)
    except OSError as why:                       # no room for the copy (it is as large as the pileup): this sample's error, said plainly
        raise devmod.PileupIOError("cannot write the copy of %s with escaped contig names: %s (TMPDIR names another place)" % (plan.pileup_path, why))
    kept = plan.excluded
    try:
        if plan.excluded is not None:
            plan.excluded = (utf8_names.escape_names(plan.excluded[0]),) + tuple(plan.excluded[1:])
        _call_one(plan, dev, params, (utf8_names.escape_names(snp_arrays[0]),) + tuple(snp_arrays[1:]))
        if plan.vcf_path:
            utf8_names.unescape_vcf_chrom(plan.vcf_path)
    finally:
        os.unlink(plan.read_path)
        plan.read_path, plan.excluded = plan.pileup_path, kept


def _call_one(plan, dev, params, snp_arrays):
    """The device part of call_consensus for one sample: site set, streamed call, the checks the reference's loop makes, outputs."""
    args = plan.args
    ss, snp_slots, _ = build_siteset(dev, snp_arrays, plan.excluded)
    try:
        _call_one_on(plan, dev, params, ss, snp_slots)
    finally:
        ss.close()                                # (the batch command comes here once per sample with non-ASCII names: no site set left behind)


def _call_one_on(plan, dev, params, ss, snp_slots):
    args = plan.args
    timing.mark("site set")
    results, rcs, _ = dev.call_consensus_files(ss, [plan.read_path], params, want_counts=True, want_line_offsets=True,
                                               want_depth_sum=bool(getattr(args, "amdMetricsRefFasta", None)))
    timing.mark("streamed call")
    all_pos = bool(args.vcfAllPos and plan.vcf_path)
    try:
        # (--vcfAllPos: the Records of ALL lines are checked, in file order, by the all-lines pass — of _write_outputs, or here and now
        # when the scan has already met a line it cannot take: which line ends the run is decided among all of them)
        if all_pos and int(rcs[0]) in (L.E_PILEUP, L.E_UNSUPPORTED):
            dev.call_all_lines(ss, plan.read_path, params, capacity=results[0].n_lines, check=True)
        if all_pos:
            dev.raise_file_status(plan.read_path, int(rcs[0]), results[0], check=False)
        else:
            dev.raise_file_errors(ss, plan.read_path, params, int(rcs[0]), results[0])
    except devmod.PileupFormatError as err:
        _raise_as_reference(err, plan.read_path, bool(args.vcfAllPos))
    _write_outputs(plan, dev, ss, snp_slots, results[0])


def call_consensus_batch(args):
    """``cfsan_snp_pipeline call_consensus_batch`` — an extension of this build, not a reference subcommand: the
    call_consensus step of run.py:704-718 for every sample directory listed in sampleDirsFile, in one process.  The
    options are call_consensus's; ``-o`` / ``-e`` / ``--pileupName`` are file NAMES inside each sample directory (as the
    ``{1}/...`` templates of run.py).  Samples are dealt round-robin to the visible GPUs, one host thread and one
    stream of files per GPU.  A failing sample is reported as a sample error and the others still run."""
    utils.print_log_header()
    utils.print_arguments(args)
    sample_directories_list_path = args.sampleDirsFile
    if utils.verify_non_empty_input_files("File of sample directories", [sample_directories_list_path]) > 0:
        utils.global_error(None)
    with open(sample_directories_list_path, "r") as f:
        sample_dirs = [line.rstrip() for line in f]
    sample_dirs = [d for d in sample_dirs if d]
    snp_list_file_path = args.snpListFile
    if utils.verify_existing_input_files("Snplist file", [snp_list_file_path]) > 0:
        utils.global_error("Error: cannot call consensus without the snplist file.")

    plans, failed = [], 0
    for d in sample_dirs:
        plan = _Plan(args, os.path.join(d, args.pileupName), os.path.join(d, args.consensusFile),
                     os.path.join(d, args.excludeFile) if args.excludeFile else None)
        plan.batch = True
        if utils.verify_non_empty_input_files("Pileup file", [plan.pileup_path]) > 0:
            utils.sample_error("Error: cannot call consensus without the pileup file.", continue_possible=True)
            failed += 1
            continue
        source_files = [snp_list_file_path, plan.pileup_path]
        if plan.exclude_path:
            if utils.verify_existing_input_files("Exclude file", [plan.exclude_path]) > 0:
                utils.sample_error("Error: cannot call consensus without the file of excluded positions.", continue_possible=True)
                failed += 1
                continue
            plan.excluded = utils.read_vcf_site_arrays(plan.exclude_path)
            source_files.append(plan.exclude_path)
        if not args.forceFlag and not utils.target_needs_rebuild(source_files, plan.consensus_path):
            utils.verbose_print("Consensus call file %s has already been freshly built.  Use the -f option to force a rebuild." % plan.consensus_path)
            continue
        plans.append(plan)
    if not plans:
        return

    snp_arrays = utils.read_snp_position_arrays(snp_list_file_path)
    utils.verbose_print("snp position list length = %d" % len(snp_arrays[2]))
    params = devmod.make_params(args.minBaseQual, args.minConsFreq, args.minConsDpth, args.minConsStrdDpth, args.minConsStrdBias)
    pinned = os.environ.get("SNPGPU_DEVICE", os.environ.get("LOCAL_RANK"))
    devices = [int(pinned)] if pinned is not None else list(range(max(1, devmod.device_count())))
    devices = devices[:max(1, len(plans))]
    errors = []                                   # (plan, exception) of samples that failed on the device
    lock = threading.Lock()

    def worker(dev_index, my_plans):
        n_done = 0                                # the plans from here on are reported as failed when the worker dies
        dev = None
        writer = None
        try:
            dev = devmod.Device(dev_index)
            # ONE site set for all samples of this GPU: the snplist plus every sample's exclude positions, with the flags of the
            # snplist only; a file's own exclude list travels with the file (per-file slots), so the samples of step 7.2 — each
            # with its own var.flt_removed.vcf — go through the device as one stream of files like those of step 7.1
            lists = [snp_arrays + (L.SITE_IN_SNPLIST,)] + [p.excluded + (0,) for p in my_plans if p.excluded is not None]
            ss, slots = dev.siteset_from_lists(lists)
            snp_slots = slots[0]
            excl_slots, k = {}, 1
            for p in my_plans:
                if p.excluded is not None:
                    excl_slots[id(p)] = slots[k]
                    k += 1
            with_excl = bool(excl_slots)
            # a stream of at most `step` files per library call: the per-site records of a call are files x sites x 138 bytes
            # on the host (0.4 GB for 16 files x 200 000 sites)
            # (16 files per call at most: the output files of one call are written by a thread of their own while the next
            # call streams — text formatting and file writes of 16 samples take about as long as their pileups take to arrive)
            step = max(1, min(16, (1 << 29) // max(1, 138 * len(ss))))

            def write_part(items):
                for plan, res, flags in items:
                    try:
                        with lock:            # the log lines of one sample stay together
                            _write_outputs(plan, dev, ss, snp_slots, res, flags)
                    except Exception as err:  # noqa: B902  (reported per sample below)
                        with lock:
                            errors.append((plan, err))

            for k0 in range(0, len(my_plans), step):
                part = my_plans[k0:k0 + step]
                exclude = [excl_slots.get(id(p), np.zeros(0, np.int64)) for p in part] if with_excl else None
                results, rcs, _ = dev.call_consensus_files(ss, [p.pileup_path for p in part], params, want_counts=True,
                                                           want_line_offsets=True, exclude=exclude,
                                                           want_depth_sum=bool(getattr(args, "amdMetricsRefFasta", None)))
                if writer is not None:
                    writer.join()             # one writer at a time: the device context below is this thread's again
                    writer = None
                in_background = []
                for plan, rc, res in zip(part, rcs, results):
                    try:
                        flags = ss.flags
                        if with_excl:
                            flags = ss.flags.copy()
                            mine = excl_slots.get(id(plan))
                            if mine is not None:
                                flags[mine[mine >= 0]] |= L.SITE_EXCLUDED
                        # (positions that are only on OTHER samples' exclude lists are nothing this sample is asked about)
                        dev.raise_file_errors(ss, plan.pileup_path, params, int(rc), res, wanted=(flags != 0) if with_excl else None)
                        needs_device = bool(plan.vcf_path) and (args.vcfAllPos or res.n_matched > int(np.count_nonzero(res.line_offsets)))
                        if needs_device:      # (the all-lines pass: a device call, so here and now)
                            with lock:
                                _write_outputs(plan, dev, ss, snp_slots, res, flags)
                        else:
                            in_background.append((plan, res, flags))
                    except Exception as err:  # noqa: B902  (reported per sample below)
                        if getattr(err, "scan_code", 0) == 3:
                            # bytes >= 0x80: contig names that are not plain ASCII go through the escaped copy, alone (utf8_names.py)
                            try:
                                with lock:
                                    _call_one_with_escaped_names(plan, dev, params, snp_arrays, err)
                                err = None
                            except Exception as err2:                # noqa: B902
                                err = err2
                        if err is not None:
                            with lock:
                                errors.append((plan, err))
                    n_done += 1
                if in_background:
                    writer = threading.Thread(target=write_part, args=(in_background,))
                    writer.start()
            if writer is not None:
                writer.join()
            ss.close()
        except Exception as err:              # noqa: B902 — the device, the site set or a whole call failed: every sample left is reported
            with lock:
                errors.extend((plan, err) for plan in my_plans[n_done:])
        finally:
            if writer is not None:
                writer.join()                 # (its samples' files are complete before the device goes)
            if dev is not None:
                dev.close()

    threads = [threading.Thread(target=worker, args=(dv, plans[i::len(devices)])) for i, dv in enumerate(devices)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for plan, err in errors:
        kind = getattr(err, "reference_exception", None) or type(err)      # (a malformed pileup: the class the reference ends with)
        utils.sample_error("Error: call_consensus failed for sample %s: %s: %s" % (plan.sample_name, kind.__name__, err),
                           continue_possible=True)
    if failed or errors:
        utils.verbose_print("%d of %d samples failed." % (failed + len(errors), len(sample_dirs)))

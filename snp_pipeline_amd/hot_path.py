"""hot_path_batch: steps 4 (site calling) to 11 of the pipeline as ONE job — every pileup crosses the host link once.

The reference runs these steps as separate process arrays over a shared file system (run.py:662-784): call_sites
(call_sites.py:89-108), filter_regions, merge_sites twice, call_consensus twice per sample (run.py:704-710 and :712-718),
snp_matrix twice, snp_reference twice, distance twice — and every one of call_sites and the two call_consensus passes reads the
sample's reads.all.pileup (0.4 GB per 5 Mbp x 30x sample) again.  This module is the same chain of steps, writing the same
files with the same bytes, arranged around the device instead of around the file system:

  1  every rank (= one GPU; torchrun-able) streams the pileups of ITS samples (a contiguous block of the sorted sample
     directories) into HBM and KEEPS them (``Device.pileups``; files past the memory budget are re-streamed in step 3);
     site calling (csrc/varscan.hip) runs on each file while the next one arrives; host threads finish each sample as its
     records come back: var.flt.vcf
  2  C1: all-gather of every sample's (CHROM, POS) records; dense-region filter (K3), both site unions (K4) on every rank —
     identical results everywhere; rank 0 writes snplist.txt / snplist_preserved.txt and the two filtered directory lists,
     every rank the var.flt_preserved.vcf / var.flt_removed.vcf of its samples
  3  ONE scan + call over the resident pileups at the positions of snplist.txt, with per-site records; the preserved flow
     (snplist_preserved.txt columns, ``Region`` for the sample's removed positions) is derived from it on the device
     (csrc/flows.hip); consensus.fasta / consensus.vcf / consensus_preserved.fasta / consensus_preserved.vcf of a group of
     samples are written by host threads while the next group is on the device
  4  4-bit pack of both matrices, C2 all-gather of the rows, all-pairs distance tiles dealt to the ranks, row-band exchange;
     snpma*.fasta (every rank writes its block of the file), the four TSVs and referenceSNP*.fasta (rank 0)

Options of the individual steps are given as the reference gives them: the ``*_ExtraParams`` strings (argument or the
environment variable of the same name), parsed by the step's own argument parser.
"""
from __future__ import print_function

import argparse
import concurrent.futures
import ctypes
import os
import shlex
import threading
import time

import numpy as np

from . import _lib as L
from . import device as devmod
from . import filter_regions as fr
from . import merge_sites as ms
from . import snp_reference
from . import utils
from . import varscan
from . import vcf_writer
from .utils import verbose_print

INGEST_BATCH = 128          # files per streamed ingest call (the record arrays of a call are files x capacity x 48 bytes)
VARSCAN_CAPACITY = 16384    # records per file in those arrays; a file with more is repeated alone


class _Sample(object):
    __slots__ = ("index", "dir", "name", "pileup", "ok", "error", "store_index", "vcf_lines", "header", "sites", "removed",
                 "n_lines", "n_rows")

    def __init__(self, index, sample_dir, pileup_name):
        self.index = index                      # position in the sorted list of sample directories
        self.dir = sample_dir
        self.name = os.path.basename(os.path.abspath(sample_dir))   # basename(dirname(<dir>/file)), as every step derives it
        self.pileup = os.path.join(sample_dir, pileup_name)
        self.ok = True
        self.error = None
        self.store_index = -1
        self.vcf_lines = self.header = self.sites = self.removed = None
        self.n_lines = self.n_rows = 0


def _step_args(step, fixed, extra):
    """The Namespace the step's own parser makes of its ExtraParams string (flags, defaults and validators are the CLI's)."""
    from . import cfsan_snp_pipeline as cli
    return cli.parse_argument_list([step] + shlex.split(extra or "") + fixed)


class _Comm(object):
    """torch.distributed when the job has more than one rank (backend nccl = RCCL over xGMI; gloo in the one-GPU tests)."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.one_gpu = os.environ.get("SNPGPU_PIPELINE_ONE_GPU") == "1"      # functional tests: all ranks on device 0 over gloo
        self.dist = None
        self.owns_group = False
        if self.one_gpu:
            self.local_rank = 0
        if self.world > 1:
            import torch
            import torch.distributed as dist
            torch.cuda.set_device(self.local_rank)
            if not dist.is_initialized():
                if self.one_gpu:
                    dist.init_process_group("gloo")
                else:
                    dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
                self.owns_group = True
            self.dist = dist

    def close(self):
        """Leave the process group this job created (its helper threads must not outlive the interpreter's teardown)."""
        if self.dist and self.owns_group and self.dist.is_initialized():
            self.dist.destroy_process_group()
            self.owns_group = False

    def barrier(self):
        if self.dist:
            import torch
            torch.cuda.synchronize()
            self.dist.barrier()

    def gather_objects(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank (small things only: names, sizes, error counts)."""
        if not self.dist:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out


def hot_path_batch(args):
    """Entry point of ``cfsan_snp_pipeline hot_path_batch`` (an extension of this build; see the module docstring)."""
    utils.print_log_header(classpath=True)
    utils.print_arguments(args)
    t_start = time.perf_counter()
    import torch                                   # device tensors, pinned host buffers and the collectives: plumbing
    from . import sharding
    comm = _Comm()
    rank, world = comm.rank, comm.world
    timings = {}

    def lap(name, t0):
        timings[name] = timings.get(name, 0.0) + time.perf_counter() - t0

    # ---- arguments: what run.py:662-784 puts on the command lines of the steps -------------------------------------------
    dirs_file = args.sampleDirsFile
    ref_path = args.referenceFile
    if utils.verify_non_empty_input_files("File of sample directories", [dirs_file]) > 0:
        utils.global_error(None)
    utils.verify_non_empty_input_files("Reference file", [ref_path], error_handler="global")
    with open(dirs_file, "r") as f:
        unsorted_dirs = [d for d in (line.rstrip() for line in f) if d]
    sorted_dirs = sorted(unsorted_dirs)
    work_dir = args.workDir or os.path.dirname(os.path.abspath(dirs_file))

    def env(name, given):
        return given if given is not None else (os.environ.get(name) or "")

    fr_args = _step_args("filter_regions", ["-n", "var.flt.vcf", dirs_file, ref_path], env("FilterRegions_ExtraParams", args.filterRegionsExtraParams))
    ms_args = _step_args("merge_sites", [dirs_file, dirs_file + ".OrigVCF.filtered"], env("MergeSites_ExtraParams", args.mergeSitesExtraParams))
    cc_extra = env("CallConsensus_ExtraParams", args.callConsensusExtraParams)
    cc_args = _step_args("call_consensus", ["--vcfRefName", os.path.basename(ref_path), "--vcfFileName", "consensus.vcf", "x.pileup"], cc_extra)
    vs_opts = varscan.Options(env("VarscanMpileup2snp_ExtraParams", args.varscanExtraParams))
    want_vcf = not args.noConsensusVcf
    # --vcfAllPos (a row for every line of the pileup, call_consensus.py:148-151) is the per-sample command's all-lines pass: the job
    # writes the FASTA files and everything downstream, and lets that command write every sample's two VCF files at the end
    vcf_all_pos = bool(cc_args.vcfAllPos and want_vcf)
    if vcf_all_pos:
        want_vcf = False
    outputs = {k: os.path.join(work_dir, v) for k, v in (
        ("snplist", "snplist.txt"), ("snplist_p", "snplist_preserved.txt"), ("snpma", "snpma.fasta"), ("snpma_p", "snpma_preserved.fasta"),
        ("pairs", "snp_distance_pairwise.tsv"), ("matrix", "snp_distance_matrix.tsv"), ("pairs_p", "snp_distance_pairwise_preserved.tsv"),
        ("matrix_p", "snp_distance_matrix_preserved.tsv"), ("refsnp", "referenceSNP.fasta"), ("refsnp_p", "referenceSNP_preserved.fasta"))}
    filtered1, filtered2 = dirs_file + ".OrigVCF.filtered", dirs_file + ".PresVCF.filtered"

    samples = [_Sample(i, d, args.pileupName) for i, d in enumerate(sorted_dirs)]
    n_total = len(samples)
    lo, hi = sharding.shard_bounds(n_total, rank, world)
    mine = samples[lo:hi]

    # make-style freshness for the job as a whole: every top-level output newer than every input
    if not args.forceFlag:
        inputs = [dirs_file, ref_path] + [s.pileup for s in samples]
        per_sample = [os.path.join(s.dir, n) for s in samples for n in ("var.flt.vcf", "consensus.fasta", "consensus_preserved.fasta")]
        if all(not utils.target_needs_rebuild(inputs, t) for t in list(outputs.values()) + per_sample):
            verbose_print("All outputs of the hot path have already been freshly built.  Use the -f option to force a rebuild.")
            return

    failed = 0
    for s in mine:
        if utils.verify_non_empty_input_files("Pileup file", [s.pileup]) > 0:
            s.ok, s.error = False, "Error: cannot process sample %s without its pileup file." % s.name

    dev = devmod.Device(comm.local_rank)
    dev.use_torch_stream()
    torch.cuda.set_device(comm.local_rank)
    store = dev.pileups(int(args.residentBytes or 0))
    try:
        # host memory for the per-site results of step 3, allocated AND touched while the pileups stream in.  Plain pageable
        # memory: on this platform a device-to-host copy into touched pageable memory runs at the pinned rate (55 GB/s), while
        # pinning a gigabyte takes 0.23 s during which every other thread's copies stand still (tools/probe/pin_probe.cpp)
        group_bytes = int(args.groupBytes) if args.groupBytes else (1 << 30)
        arenas = [None, None]

        def alloc_arenas():
            for k in range(2 if hi - lo > 1 else 1):
                a = np.empty(group_bytes, dtype=np.uint8)
                ctypes.memset(a.ctypes.data, 0, a.nbytes)      # touches the pages with the GIL released (ndarray.fill would hold it)
                arenas[k] = torch.from_numpy(a)

        arena_thread = threading.Thread(target=alloc_arenas)
        arena_thread.start()
        # ================================ 1: pileups -> HBM, site calling -> var.flt.vcf ===================================
        t0 = time.perf_counter()
        todo = [s for s in mine if s.ok]
        vparams = vs_opts.device_params()
        pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(2, min(16, (os.cpu_count() or 4) // 4)))
        redo = []

        def finish_sample(s, records, n_lines):
            """Host half of call_sites for one sample (Fisher's exact test, VCF text: csrc/varscan_rows.hip), then the records
            filter_regions / merge_sites read from the file it has just written."""
            try:
                vcf_path = os.path.join(s.dir, "var.flt.vcf")
                s.n_lines = n_lines
                s.n_rows = varscan._write_vcf(vcf_path, s.pileup, records, vs_opts)
                s.header, s.vcf_lines, s.sites = fr._read_vcf(vcf_path)
            except Exception as err:                             # noqa: B902 — reported as this sample's error
                s.ok, s.error = False, "Error: call_sites failed for sample %s: %s: %s" % (s.name, type(err).__name__, err)

        for b0 in range(0, len(todo), INGEST_BATCH):
            batch = todo[b0:b0 + INGEST_BATCH]
            bufs = store.ingest_buffers(len(batch), VARSCAN_CAPACITY)
            sites, counts, status, rcs, done = bufs
            first = len(store)
            failure = []

            def run_ingest(batch=batch, bufs=bufs):
                try:
                    store.ingest_into([s.pileup for s in batch], vparams, VARSCAN_CAPACITY, bufs)
                except BaseException as err:                     # noqa: B902 — handed to the main thread
                    failure.append(err)
                    bufs[4][:] = 1

            t_call = time.perf_counter()
            th = threading.Thread(target=run_ingest)
            th.start()
            pending = list(range(len(batch)))
            futures = []
            while pending:
                still = []
                for k in pending:
                    if not done[k]:
                        still.append(k)
                        continue
                    s = batch[k]
                    s.store_index = first + k
                    if failure:
                        continue
                    if rcs[k] == L.E_IO:
                        s.ok, s.error = False, "Error: cannot open or read the pileup file %s" % s.pileup
                    elif rcs[k] == L.E_PILEUP:
                        s.ok, s.error = False, "Error: call_sites failed for sample %s: ValueError: Invalid format for pileup at byte %d of %s" % (
                            s.name, int(status[k, 0]), s.pileup)
                    elif rcs[k] != 0:
                        s.ok, s.error = False, "Error: call_sites failed for sample %s (device error %d)" % (s.name, int(rcs[k]))
                    elif counts[k] > VARSCAN_CAPACITY:
                        redo.append(s)                           # more records than the shared array holds: alone, afterwards
                    else:
                        futures.append(pool.submit(finish_sample, s, sites[k, :counts[k]], int(status[k, 1])))
                pending = still
                if pending:
                    time.sleep(0.0005)
            th.join()
            lap("1a   of which: streamed ingest calls", t_call)
            if failure:
                raise failure[0]
            t_tail = time.perf_counter()
            for fu in futures:
                fu.result()
            lap("1b   of which: waiting for the last var.flt.vcf files", t_tail)
        for s in redo:
            ptr, nbytes = store.get(s.store_index)
            try:
                records, n_lines = dev.varscan_dev(ptr, nbytes, vparams, capacity=4 * VARSCAN_CAPACITY) if ptr else dev.varscan_file(s.pileup, vparams)
                finish_sample(s, records, n_lines)
            except Exception as err:                             # noqa: B902
                s.ok, s.error = False, "Error: call_sites failed for sample %s: %s: %s" % (s.name, type(err).__name__, err)
        pool.shutdown()
        lap("1 ingest + site calling", t0)

        # ================================ 2: C1 + filter_regions + merge_sites x 2 =========================================
        t0 = time.perf_counter()
        # every rank learns every sample's records: contig names as objects (a few strings), the records as one all-gather
        local_names = sorted({c for s in mine if s.ok for c in s.sites[0]})
        contigs = sorted({c for names in comm.gather_objects(local_names) for c in names})
        cid = {c: i for i, c in enumerate(contigs)}
        rec_count = np.zeros(hi - lo, dtype=np.int64)
        keys_local = []
        for k, s in enumerate(mine):
            if not s.ok:
                rec_count[k] = -1                                # no var.flt.vcf: the steps below report it missing
                continue
            names, cidx, pos = s.sites
            if len(pos) and (pos.min() < 0 or pos.max() >= (1 << 32)):
                raise ValueError("VCF position out of range")    # as merge_sites (utils.py:1127 has no such record either)
            lut = np.asarray([cid[c] for c in names] + [0], dtype=np.int64)
            keys_local.append((lut[cidx.astype(np.int64)] << 32) | pos)
            rec_count[k] = len(pos)
        keys_local = np.concatenate(keys_local) if keys_local else np.zeros(0, np.int64)
        if world > 1:
            dv = "cpu" if comm.one_gpu else "cuda"
            all_keys, _ = sharding.all_gather_varlen(torch.from_numpy(keys_local).to(dv))
            all_cnt, _ = sharding.all_gather_varlen(torch.from_numpy(rec_count).to(dv))
            all_keys, all_cnt = all_keys.cpu().numpy(), all_cnt.cpu().numpy()
        else:
            all_keys, all_cnt = keys_local, rec_count
        has_vcf = all_cnt >= 0
        cnt0 = np.maximum(all_cnt, 0)
        rec_off = np.zeros(n_total + 1, dtype=np.int64)
        np.cumsum(cnt0, out=rec_off[1:])
        rec_sample = np.repeat(np.arange(n_total, dtype=np.int64), cnt0)
        rec_cid = (all_keys >> 32).astype(np.uint32)
        rec_pos = all_keys & 0xFFFFFFFF
        n_bad = int((~has_vcf).sum())
        if rank == 0:
            if n_bad == n_total:
                utils.global_error("Error: all %d VCF files were missing or empty." % n_bad)
            elif n_bad > 0:
                utils.sample_error("Error: %d VCF files were missing or empty." % n_bad, continue_possible=True)

        def unique_per_sample(keep):
            """Distinct (contig, position) pairs per sample among the records `keep` selects (the size of merge_sites' snp_set)."""
            pairs = np.unique(np.stack([rec_sample[keep], all_keys[keep]]), axis=1)
            return np.bincount(pairs[0], minlength=n_total)

        def site_union(keep, which):
            """merge_sites.py:91-117 over the records `keep`: --maxsnps sample exclusion, then the union with its carriers."""
            excluded = np.zeros(n_total, bool)
            if ms_args.maxSnps >= 0:
                per = unique_per_sample(keep)
                excluded = has_vcf & (per > ms_args.maxSnps)
                if rank == 0:
                    for i in np.flatnonzero(excluded):
                        verbose_print("Excluding sample %s having %d snps." % (samples[i].name, per[i]))
            inc = has_vcf & ~excluded
            carrier_ids = np.flatnonzero(inc)                    # carriers are indices into the INCLUDED samples, sorted-dir order
            remap = np.full(n_total, -1, dtype=np.int64)
            remap[carrier_ids] = np.arange(len(carrier_ids))
            use = keep & inc[rec_sample]
            if use.any():
                uniq, off, car = dev.merge_sites(all_keys[use].astype(np.uint64), remap[rec_sample[use]].astype(np.uint32))
            else:
                uniq, off, car = np.zeros(0, np.uint64), np.zeros(1, np.uint32), np.zeros(0, np.uint32)
            if rank == 0:
                verbose_print("Found %d snp positions across %d sample vcf files." % (len(uniq), n_total))
                ms.write_snplist(outputs[which], contigs, uniq, off, car, [samples[i].name for i in carrier_ids])
                with open(filtered1 if which == "snplist" else filtered2, "w") as f:
                    for d in unsorted_dirs:                      # original order (merge_sites.py:127-131)
                        if not excluded[dir_index[d]]:
                            f.write("%s\n" % d)
            return uniq.astype(np.int64), excluded

        dir_index = {d: i for i, d in enumerate(sorted_dirs)}
        every = np.ones(len(all_keys), dtype=bool)
        lap("2a   of which: gathering the records", t0)
        t2 = time.perf_counter()
        list1, excluded1 = site_union(every, "snplist")
        lap("2b   of which: site union + snplist.txt", t2)
        t2 = time.perf_counter()

        # filter_regions (filter_regions.py:205-383): dense windows + contig edges -> merged bad regions -> classification of
        # every record; mode all unions the regions over the samples, mode each keeps them per sample; outgroup samples bypass
        outgroup = set()
        if fr_args.outGroupFile is not None:
            if utils.verify_non_empty_input_files("File of outgroup samples", [fr_args.outGroupFile]) > 0:
                utils.global_error(None)
            with open(fr_args.outGroupFile, "r") as f:
                outgroup = {line.rstrip() for line in f}
        try:
            contig_lengths = utils.read_fasta_lengths(ref_path)
        except (IOError, OSError, UnicodeDecodeError):
            utils.global_error("Error: cannot open the reference fastq file, or fail to read the contigs in the reference fastq file.")
        is_out = np.asarray([s.name in outgroup for s in samples], dtype=bool)
        filt = has_vcf & ~is_out                                 # the samples that take part in the region step
        filt_ids = np.flatnonzero(filt)
        part_rank = np.full(n_total, -1, dtype=np.int64)
        part_rank[filt_ids] = np.arange(len(filt_ids))
        takes_part = filt[rec_sample]
        lap("2c   of which: (records laid out)", t2)
        t2 = time.perf_counter()
        removed = np.zeros(len(all_keys), dtype=bool)
        removed[takes_part] = fr.removed_flags(dev, contigs, contig_lengths, part_rank[rec_sample[takes_part]], rec_cid[takes_part], rec_pos[takes_part],
                                               len(filt_ids), fr_args.edgeLength, fr_args.maxSnpsList, fr_args.windowSizeList, per_sample=fr_args.mode == "each")
        preserved = every & ~removed
        lap("2d   of which: dense windows, region merge, classification", t2)
        t2 = time.perf_counter()
        list2, excluded2 = site_union(preserved, "snplist_p")
        lap("2e   of which: preserved site union + snplist_preserved.txt", t2)
        t2 = time.perf_counter()
        # the split VCF files of this rank's samples: written by host threads while step 3 keeps the device busy
        split_pool = concurrent.futures.ThreadPoolExecutor(max_workers=4)
        split_files = []
        for s in mine:
            if not s.ok:
                continue
            vcf_path = os.path.join(s.dir, "var.flt.vcf")
            if is_out[s.index]:
                split_files.append(split_pool.submit(fr.write_outgroup_preserved_and_removed_vcf_files, vcf_path, s.header))
            else:
                s.removed = removed[rec_off[s.index]:rec_off[s.index + 1]]
                split_files.append(split_pool.submit(fr.write_preserved_and_removed_vcf_files, vcf_path, s.header, s.vcf_lines, s.removed))
        lap("2f   of which: var.flt_preserved / _removed.vcf files handed to writer threads", t2)
        lap("2 site union + region filter", t0)

        # ================================ 3: both consensus flows from one scan + call =====================================
        t0 = time.perf_counter()
        prm = devmod.make_params(cc_args.minBaseQual, cc_args.minConsFreq, cc_args.minConsDpth, cc_args.minConsStrdDpth, cc_args.minConsStrdBias)
        # the site set: snplist.txt, plus removed positions of this rank's samples that are not in it (a sample merge_sites
        # excluded for --maxsnps is still called, run.py:704-718, and its exclude list is parsed: call_consensus.py:147-151),
        # plus what snplist_preserved.txt has and snplist.txt has not: with --maxsnps a sample can be out of the first list for its
        # var.flt.vcf and in the second for its shorter var.flt_preserved.vcf (found by tools/fuzz_jobs.py)
        own_removed = {s.index: (all_keys[rec_off[s.index]:rec_off[s.index + 1]][s.removed] if (s.ok and s.removed is not None) else np.zeros(0, np.int64))
                       for s in mine}
        extra = np.setdiff1d(np.concatenate(list(own_removed.values())) if own_removed else np.zeros(0, np.int64), list1)
        set_keys = list1
        for more in (extra, np.setdiff1d(list2, list1)):
            if len(more):
                set_keys = np.union1d(set_keys, more)
        S = len(set_keys)
        cols1 = np.searchsorted(set_keys, list1).astype(np.uint32)
        cols2 = np.searchsorted(set_keys, list2).astype(np.uint32)
        in1 = np.zeros(S, dtype=np.uint8)
        in1[cols1] = 1
        in2 = np.zeros(S, dtype=np.uint8)
        in2[cols2] = 1
        col_of2 = np.full(S, -1, dtype=np.int32)
        col_of2[cols2] = np.arange(len(cols2), dtype=np.int32)
        ss = devmod.SiteSet.from_arrays(dev, [c.encode("utf-8") for c in contigs], set_keys.astype(np.uint64), in1 * np.uint8(L.SITE_IN_SNPLIST))
        lap("3-   of which: site set", t0)
        identity1 = len(set_keys) == len(list1)
        S1, S2 = len(list1), len(list2)
        # collect_metrics by-products (call_consensus --amdMetricsRefFasta, given through CallConsensus_ExtraParams): the depth
        # column is summed by the same scan, the gaps are counted in the rows that are written anyway
        metrics_ref = getattr(cc_args, "amdMetricsRefFasta", None)
        metrics_ref_len = sum(utils.read_fasta_lengths(metrics_ref).values()) if metrics_ref else 0
        filters_desc = vcf_writer.filter_descriptions(cc_args.minConsFreq, cc_args.minConsDpth, cc_args.minConsStrdDpth, cc_args.minConsStrdBias)
        filter_names = [n for n, _ in filters_desc]
        d_cols1, d_cols2 = torch.from_numpy(cols1.astype(np.int32)).cuda(), torch.from_numpy(cols2.astype(np.int32)).cuda()
        d_col_of2 = torch.from_numpy(col_of2).cuda()
        d_col_of1 = torch.from_numpy(np.where(in1 != 0, np.cumsum(in1, dtype=np.int64) - 1, -1).astype(np.int32)).cuda()
        d_err = torch.zeros(4, dtype=torch.int32, device="cuda")
        callable_ = [s for s in mine if s.ok]
        n_local = len(callable_)
        rows1 = torch.full((max(n_local, 1), max(S1, 1)), 0x2D, dtype=torch.uint8, device="cuda")     # the consensus rows, for step 4
        rows2 = torch.full((max(n_local, 1), max(S2, 1)), 0x2D, dtype=torch.uint8, device="cuda")
        row_ok = np.zeros(n_local, dtype=bool)
        writer = concurrent.futures.ThreadPoolExecutor(max_workers=1)
        per_sample_bytes = max(S1, 1) + max(S2, 1) + 2 * max(S, 1) + 8 * max(S, 1) + 32 + (128 * max(S, 1) if want_vcf else 0) + 64
        group = max(1, min(256, group_bytes // per_sample_bytes))
        if n_local >= 32:
            group = min(group, (n_local + 1) // 2)             # at least two groups: the files of one are written while the next is on the device
        g_alloc = min(group, max(n_local, 1))
        arena_thread.join()
        h2d_extra = [0]

        def host_set(k):
            """Result arrays of one group, carved out of arena k (allocated while the pileups were streaming in)."""
            need = g_alloc * per_sample_bytes + 4096
            if arenas[k] is None or arenas[k].numel() < need:
                arenas[k] = torch.from_numpy(np.zeros(need, dtype=np.uint8))
            at = [0]

            def carve(shape, dtype):
                nbytes = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
                start = (at[0] + 63) // 64 * 64
                at[0] = start + nbytes
                return arenas[k][start:start + nbytes].view(dtype).view(*shape)

            return {"g": g_alloc,
                    "status": carve((g_alloc, 4), torch.int64),
                    "line": carve((g_alloc, max(S, 1)), torch.int64),
                    "counts": carve((g_alloc, max(S, 1), 128), torch.uint8) if want_vcf else None,
                    "base1": carve((g_alloc, max(S1, 1)), torch.uint8),
                    "base2": carve((g_alloc, max(S2, 1)), torch.uint8),
                    "filt1": carve((g_alloc, max(S, 1)), torch.uint8),
                    "filt2": carve((g_alloc, max(S, 1)), torch.uint8)}

        host_sets = [None, None]
        d_base = torch.empty((g_alloc, max(S, 1)), dtype=torch.uint8, device="cuda")
        d_filt = torch.empty((g_alloc, max(S, 1)), dtype=torch.uint8, device="cuda")
        d_filt2 = torch.empty((g_alloc, max(S, 1)), dtype=torch.uint8, device="cuda")
        d_line = torch.zeros((g_alloc, max(S, 1)), dtype=torch.int64, device="cuda")
        d_status = torch.empty((g_alloc, 4), dtype=torch.int64, device="cuda")
        d_counts = torch.empty((g_alloc, max(S, 1), 128), dtype=torch.uint8, device="cuda") if want_vcf else None
        vcf_date = None
        pending_write = None
        vcf_again = []                                           # samples whose VCF files the per-sample command writes at the end (see the status checks)
        vcf_later = set()                                        # ... of them, those whose group must not write a VCF meanwhile

        def write_group(part, hs, g0, spill=None):
            """FASTA + VCF files of one group, both flows (host threads inside the library); returns the samples that failed."""
            jobs, owners = [], []
            counts_np = hs["counts"].numpy().view(devmod.COUNTS_DTYPE).reshape(hs["g"], max(S, 1)) if want_vcf else None
            for k, s in enumerate(part):
                if not s.ok:
                    continue
                for flow in (1, 2):
                    seq = (hs["base1"] if flow == 1 else hs["base2"]).numpy()[k, :(S1 if flow == 1 else S2)]
                    job = {"fasta_path": os.path.join(s.dir, "consensus.fasta" if flow == 1 else "consensus_preserved.fasta"),
                           "fasta_id": s.name.encode("utf-8"), "sequence": seq}
                    if want_vcf and s.index not in vcf_later:
                        hdr = "\n".join(vcf_writer.header_lines(s.name, filters_desc, cc_args.vcfRefName, now=vcf_date)) + "\n"
                        job.update({"vcf_path": os.path.join(s.dir, "consensus.vcf" if flow == 1 else "consensus_preserved.vcf"),
                                    "vcf_header": hdr.encode("utf-8"), "counts": counts_np[k], "line_off": hs["line"].numpy()[k].view(np.uint64),
                                    "row_filters": (hs["filt1"] if flow == 1 else hs["filt2"]).numpy()[k],
                                    "site_in_flow": in1 if flow == 1 else in2})
                    jobs.append(job)
                    owners.append(s)
            res = devmod.write_consensus_files(jobs, ss, filter_names, cc_args.vcfPreserveRefCase, cc_args.vcfFailedSnpGt, n_threads=args.writerThreads,
                                               spill=spill)
            if metrics_ref:                                      # as call_consensus._record_metrics does for the two flows, in their order
                st_h = hs["status"].numpy()
                for k, s in enumerate(part):
                    if not s.ok:
                        continue
                    name = os.path.basename(cc_args.amdMetricsFile) if cc_args.amdMetricsFile else "metrics"
                    depth_sum = int(st_h[k, 3]) & 0xFFFFFFFFFFFFFFFF
                    ave = {"avePileupDepth": "%.2f" % (float(depth_sum) / float(metrics_ref_len))} if depth_sum > 0 and metrics_ref_len > 0 else {}
                    for key, row, n in (("missingPos", hs["base1"], S1), ("missingPosPreserved", hs["base2"], S2)):
                        updates = {key: str(int(np.count_nonzero(row.numpy()[k, :n] == 0x2D)))}
                        updates.update(ave)
                        utils.update_properties(os.path.join(s.dir, name), updates, keep_mtime=True)
            bad = []
            for s, job, (rc, _) in zip(owners, jobs, res):
                if rc == L.E_UNSUPPORTED:
                    bad.append((s, "Error: call_consensus failed for sample %s: ValueError: a position has more than %d distinct symbols" % (s.name, L.MAX_SYMS)))
                elif rc != 0:
                    bad.append((s, "Error: cannot write %s" % job["fasta_path"]))
            return bad

        import datetime
        vcf_date = datetime.datetime.now()
        write_failures = []
        for g0 in range(0, n_local, group):
            t_g = time.perf_counter()
            part = callable_[g0:g0 + group]
            g = len(part)
            which = (g0 // group) & 1
            if host_sets[which] is None:
                host_sets[which] = host_set(which)
            hs = host_sets[which]
            resident = [(k, s) + store.get(s.store_index) for k, s in enumerate(part)]
            res_idx = [k for k, s, ptr, _ in resident if ptr]
            if res_idx and S:
                # resident samples first in the group's arrays would need a permutation: call them in place, sample by sample
                # position, with one launch over all resident ones
                ptrs = [ptr for _, _, ptr, _ in resident if ptr]
                sizes = [n for _, _, ptr, n in resident if ptr]
                if len(res_idx) == g:
                    dev.call_consensus_many_dev(ss, ptrs, sizes, prm, d_base.data_ptr(), d_filt.data_ptr(), d_status.data_ptr(),
                                                d_counts=d_counts.data_ptr() if want_vcf else 0, d_line_off=d_line.data_ptr(),
                                                want_depth_sum=bool(metrics_ref))
                else:
                    _call_scattered(dev, ss, prm, res_idx, ptrs, sizes, d_base, d_filt, d_status, d_counts, d_line, S, want_vcf, torch,
                                    want_depth_sum=bool(metrics_ref))
            elif res_idx:
                d_status[:g] = torch.tensor([-1, 0, 0, 0], dtype=torch.int64, device="cuda")
            rest = [(k, s) for k, s, ptr, _ in resident if not ptr]
            if rest:
                # files that did not fit the memory budget: streamed again (the only pileups that cross the link twice)
                results, rcs, st = dev.call_consensus_files(ss, [s.pileup for _, s in rest], prm, want_counts=want_vcf, want_line_offsets=True,
                                                            want_depth_sum=bool(metrics_ref))
                h2d_extra[0] += int(st.bytes)
                for (k, s), rc, r in zip(rest, rcs, results):
                    if int(rc) == L.E_IO:
                        s.ok, s.error = False, "Error: cannot open or read the pileup file %s" % s.pileup
                    d_status[k] = torch.from_numpy(r.status.astype(np.int64)).cuda()
                    if S:
                        d_base[k, :S] = torch.from_numpy(r.bases).cuda()
                        d_filt[k, :S] = torch.from_numpy(r.filters).cuda()
                        d_line[k, :S] = torch.from_numpy(r.line_offsets.astype(np.int64)).cuda()
                        if want_vcf:
                            d_counts[k, :S] = torch.from_numpy(r.counts.view(np.uint8).reshape(S, 128)).cuda()
            # the preserved flow (and, when the set is wider than snplist.txt, the columns of the full flow) on the device
            excl = [np.searchsorted(set_keys, own_removed[s.index]).astype(np.uint32) for s in part]
            eoff = np.zeros(g + 1, dtype=np.int32)
            np.cumsum([len(e) for e in excl], out=eoff[1:])
            d_eoff = torch.from_numpy(eoff).cuda()
            d_eslots = torch.from_numpy(np.concatenate(excl).astype(np.int32) if eoff[-1] else np.zeros(1, np.int32)).cuda()
            d_b2 = rows2[g0:g0 + g]
            dev.region_flow_dev(d_base.data_ptr(), d_filt.data_ptr(), d_line.data_ptr(), g, S, d_cols2.data_ptr(), d_col_of2.data_ptr(), S2,
                                d_eoff.data_ptr(), d_eslots.data_ptr() if eoff[-1] else 0, d_b2.data_ptr() if S2 else 0, d_filt2.data_ptr(), d_err.data_ptr())
            if identity1:
                if S1:
                    rows1[g0:g0 + g, :S1] = d_base[:g, :S1]
            else:
                d_nofilt = torch.empty_like(d_filt2)
                d_e0 = torch.zeros(g + 1, dtype=torch.int32, device="cuda")
                dev.region_flow_dev(d_base.data_ptr(), d_filt.data_ptr(), d_line.data_ptr(), g, S, d_cols1.data_ptr(), d_col_of1.data_ptr(), S1,
                                    d_e0.data_ptr(), 0, rows1[g0:g0 + g].data_ptr() if S1 else 0, d_nofilt.data_ptr(), d_err.data_ptr())
            # results to the host
            if args.verbose >= 2:
                torch.cuda.current_stream().synchronize()
                lap("3a   of which: scan + call + flows on the device", t_g)
                t_g = time.perf_counter()
            # per sample, for the checks below: a malformed line at a listed position?  how many listed positions have a line?
            if S:
                d_bad = (d_counts[:g, :S, 23] > L.ST_OK).any(dim=1) if want_vcf else ((d_filt[:g, :S] & 0x80) != 0).any(dim=1)
                # (bytes 17-19 of a record: nonzero = the position has a record in the context's spill — more than 8 symbols, or a
                # reference field of several bytes)
                d_ovf = (d_counts[:g, :S, 17:20] != 0).any(dim=2).sum(dim=1) if want_vcf else torch.zeros(g, dtype=torch.int64, device="cuda")
                d_chk = torch.stack([d_bad.to(torch.int64), (d_line[:g, :S] != 0).sum(dim=1), d_ovf], dim=1)
            hs["status"][:g].copy_(d_status[:g], non_blocking=True)
            if S1:
                hs["base1"][:g, :S1].copy_(rows1[g0:g0 + g, :S1], non_blocking=True)
            if S2:
                hs["base2"][:g, :S2].copy_(rows2[g0:g0 + g, :S2], non_blocking=True)
            if S:
                hs["filt1"][:g, :S].copy_(d_filt[:g, :S], non_blocking=True)
                hs["filt2"][:g, :S].copy_(d_filt2[:g, :S], non_blocking=True)
                hs["line"][:g, :S].copy_(d_line[:g, :S], non_blocking=True)
                if want_vcf:
                    hs["counts"][:g, :S].copy_(d_counts[:g, :S], non_blocking=True)
            chk = d_chk.cpu().numpy() if S else np.zeros((g, 3), dtype=np.int64)      # (the stream is idle after this)
            group_spill = None
            if S and chk[:, 2].any():
                # positions with more than 8 distinct symbols: their spill records belong to ONE library call — there is one
                # when the whole group was resident; a sample of a mixed group goes back to the per-sample command
                if not rest:
                    group_spill = dev.read_symbol_spill()
                else:
                    for k, s in enumerate(part):
                        if s.ok and chk[k, 2]:                   # its consensus is as good as any; its VCF rows need the spill of a call of its own
                            vcf_again.append(s)
                            vcf_later.add(s.index)
            torch.cuda.current_stream().synchronize()
            lap("3b   of which: results to the host" if args.verbose >= 2 else "3ab  of which: device work + results to the host", t_g)
            t_g = time.perf_counter()
            # what the per-sample CLI raises for: malformed chrom / position columns anywhere, a malformed line at a listed position
            st_np = hs["status"].numpy()[:g]
            for k, s in enumerate(part):
                if not s.ok:
                    continue
                w0 = int(st_np[k, 0]) & 0xFFFFFFFFFFFFFFFF
                if w0 != 0xFFFFFFFFFFFFFFFF:
                    s.ok, s.error = False, "Error: call_consensus failed for sample %s: malformed pileup %s at byte offset %d" % (s.name, s.pileup, (w0 >> 8) - 1)
                    continue
                if S and chk[k, 0]:
                    s.ok, s.error = False, "Error: call_consensus failed for sample %s: malformed pileup line at a listed position" % s.name
                    continue
                if S and int(st_np[k, 2]) > int(chk[k, 1]):
                    # The pileup repeats a listed position.  The last line of a position is the one that counts for the consensus
                    # (call_consensus.py:171-176) and that is what the device returned; but the reference builds a Record from
                    # every such line — it ends at the first it cannot build — and writes a consensus.vcf row for each.  The
                    # all-lines pass looks at them now; the VCF files of such a sample are written by the per-sample command
                    # once the job's own files are done (a sorted pileup never comes here).
                    try:
                        _, line_flags, line_counts = dev.call_all_lines(ss, s.pileup, prm, check=False)
                        err, _ = devmod.Device.site_error(devmod.ConsensusResult(None, None, line_counts[line_flags != 0], st_np[k]))
                    except (devmod.PileupFormatError, devmod.PileupIOError) as e:
                        err = e
                    if err is not None:
                        s.ok, s.error = False, "Error: call_consensus failed for sample %s: %s" % (s.name, err)
                        continue
                    if want_vcf and s not in vcf_again:
                        vcf_again.append(s)
                row_ok[g0 + k] = True
            lap("3c   of which: status checks", t_g)
            t_g = time.perf_counter()
            if pending_write is not None:
                write_failures.extend(pending_write.result())
            pending_write = writer.submit(write_group, part, hs, g0, group_spill)
            lap("3d   of which: waiting for the previous group's files", t_g)
        t_g = time.perf_counter()
        if pending_write is not None:
            write_failures.extend(pending_write.result())
        writer.shutdown()
        lap("3e   of which: waiting for the last group's files", t_g)
        for s, msg in write_failures:
            s.ok, s.error = False, msg
            row_ok[callable_.index(s)] = False
        if int(d_err[0]) != 0:
            raise RuntimeError("an exclude slot fell outside the site set")
        t_g = time.perf_counter()
        for fu in split_files:
            fu.result()
        split_pool.shutdown()
        lap("3f   of which: waiting for the split VCF files of step 2", t_g)
        lap("3 consensus, both flows", t0)

        # ================================ 4: matrices, distances, top-level files ==========================================
        t0 = time.perf_counter()
        # which samples of the whole job have a consensus row (gathered: small)
        ref_seqs = [None]
        ok_local = np.zeros(hi - lo, dtype=bool)
        for k, s in enumerate(callable_):
            ok_local[s.index - lo] = bool(row_ok[k])
        ok_all = np.concatenate(comm.gather_objects(ok_local)) if n_total else np.zeros(0, bool)
        local_row_of = {s.index: k for k, s in enumerate(callable_)}
        for flow, rows, Sx, excluded, snpma, pairs, matrix in ((1, rows1, S1, excluded1, "snpma", "pairs", "matrix"),
                                                              (2, rows2, S2, excluded2, "snpma_p", "pairs_p", "matrix_p")):
            # snp_matrix (snp_matrix.py:79-119): the samples of the step's filtered list, sorted-dir order, that have a consensus file
            member = ok_all & ~excluded
            if rank == 0 and not member.any():
                utils.global_error("Error: all %d consensus fasta files were missing or empty." % int((~excluded).sum()))
            chunks = []
            for s in mine:
                if member[s.index]:
                    k = local_row_of[s.index]
                    chunks.append(_fasta_bytes(s.name, rows[k, :Sx].cpu().numpy() if Sx else np.zeros(0, np.uint8)))
            blob = b"".join(chunks)
            sizes = comm.gather_objects(len(blob))
            if rank == 0:
                with open(outputs[snpma], "wb") as f:
                    f.truncate(sum(sizes))
            comm.barrier()
            with open(outputs[snpma], "r+b") as f:
                f.seek(sum(sizes[:rank]))
                f.write(blob)
            # distance (distance.py:76-114): ids = sorted names, equal names keep the last record of the file
            last = {}
            for i in np.flatnonzero(member):
                last[samples[i].name] = i
            ids = sorted(last)
            order = np.asarray([last[i] for i in ids], dtype=np.int64)
            n = len(ids)
            row_bytes = dev.packed_row_bytes(Sx)
            per = (n_total + world - 1) // world if n_total else 0
            packed_local = torch.zeros((max(hi - lo, 1), row_bytes), dtype=torch.uint8, device="cuda")
            if n_local and Sx:
                tmp = torch.zeros((n_local, row_bytes), dtype=torch.uint8, device="cuda")
                dev.pack_matrix_dev(rows.data_ptr(), n_local, Sx, rows.shape[1], tmp.data_ptr())
                idx = torch.tensor([s.index - lo for s in callable_], dtype=torch.int64, device="cuda")
                packed_local[idx] = tmp
            packed_all = torch.zeros((max(world * per, 1), row_bytes), dtype=torch.uint8, device="cuda")
            sharding.all_gather_rows_into(packed_local[:hi - lo], n_total, packed_all)
            bands = sharding.RowBands(n, world)
            packed_sorted = torch.zeros((max(bands.n_padded, 1), row_bytes), dtype=torch.uint8, device="cuda")
            if n:
                packed_sorted[:n] = packed_all[torch.from_numpy(order).cuda()]
            dmat = torch.zeros((max(bands.n_padded, 1), max(bands.n_padded, 1)), dtype=torch.int32, device="cuda")
            if n and Sx:
                dev.distance_packed_dev(packed_sorted.data_ptr(), bands.n_padded, Sx, dmat.data_ptr(), rank, world)
            if world > 1 and n:
                band = bands.exchange(dmat, rank)                # the complete rows of this rank's band
                blo, bhi = bands.band_rows(rank)
                pieces = comm.gather_objects(band[:bhi - blo, :n].cpu().numpy())
                full = np.concatenate(pieces, axis=0) if rank == 0 else None
            else:
                full = dmat[:n, :n].cpu().numpy()
            if rank == 0:
                from . import distance as dmod
                dmod.write_pairwise(outputs[pairs], ids, full)
                dmod.write_matrix(outputs[matrix], ids, full)
                if ref_seqs[0] is None:
                    ref_seqs[0] = snp_reference.read_fasta_sequences(ref_path)
                snp_reference.write_reference_snp_file(ref_path, outputs["snplist" if flow == 1 else "snplist_p"],
                                                       outputs["refsnp" if flow == 1 else "refsnp_p"], match_dict=ref_seqs[0])
        comm.barrier()
        lap("4 matrices + distances", t0)
        # ---- samples whose pileup repeats a listed position: consensus.vcf rows for EVERY matching line, by the per-sample command
        #      (call_consensus.py:178-180); its consensus.fasta is the same bytes the job wrote ----------------------------------------
        if vcf_all_pos:
            vcf_again = [s for k, s in enumerate(callable_) if row_ok[k]]
        if vcf_again:
            from . import call_consensus as cc_step
            quiet = argparse.Namespace(verbose=0)
            utils.set_logging_verbosity(quiet)
            process_device, devmod._default = devmod._default, dev       # (the command's "process-wide device" is this job's context)
            try:
                for s in vcf_again:
                    if not s.ok:
                        continue
                    try:
                        for snplist, suffix, more in ((outputs["snplist"], "", []),
                                                      (outputs["snplist_p"], "_preserved", ["-e", os.path.join(s.dir, "var.flt_removed.vcf")])):
                            cc_step.call_consensus(_step_args("call_consensus", ["-f", "-l", snplist, "-o", os.path.join(s.dir, "consensus%s.fasta" % suffix),
                                                                                 "--vcfRefName", os.path.basename(ref_path), "--vcfFileName", "consensus%s.vcf" % suffix]
                                                              + more + [s.pileup], cc_extra))
                    except (Exception, SystemExit) as e:         # noqa: B902 — reported as this sample's error below
                        s.ok, s.error = False, "Error: call_consensus failed for sample %s: %s: %s" % (s.name, type(e).__name__, e)
            finally:
                devmod._default = process_device
                utils.set_logging_verbosity(args)
        st = store.stats()
        stats = {"h2d_bytes": int(st.h2d_bytes) + h2d_extra[0], "file_bytes": int(st.file_bytes), "resident_files": int(st.n_resident),
                 "files": int(st.n_files), "seconds": time.perf_counter() - t_start,
                 "ingest": {"seconds": st.seconds, "allocating": st.seconds_allocating, "waiting_for_readers": st.seconds_waiting_for_readers,
                            "waiting_for_device": st.seconds_waiting_for_device, "reader_seconds_reading": st.reader_seconds_reading,
                            "reader_seconds_waiting": st.reader_seconds_waiting, "preparing": st.seconds_preparing}, "phases": timings, "sites": S1, "sites_preserved": S2,
                 "samples": hi - lo}
        hot_path_batch.last_stats = stats
        verbose_print("# hot_path_batch rank %d: %d samples, %d pileup bytes, %d bytes copied to the device (%d files resident), %.3f s"
                      % (rank, hi - lo, stats["file_bytes"], stats["h2d_bytes"], stats["resident_files"], stats["seconds"]))
        for k in sorted(timings):
            verbose_print("#   %-34s %.3f s" % (k, timings[k]))
    finally:
        store.close()
        dev.close()
    # ---- per-sample errors: reported the way the batch subcommands do ----------------------------------------------------
    errs = [s.error for s in mine if not s.ok]
    all_errs = comm.gather_objects(errs)
    failed = sum(len(e) for e in all_errs)
    comm.close()                                             # (before anything that may end the process: sample_error exits when StopOnSampleError says so)
    for msg in errs:
        utils.sample_error(msg, continue_possible=True)
    if failed:
        verbose_print("%d of %d samples failed." % (failed, n_total))


hot_path_batch.last_stats = None


def _fasta_bytes(name, seq):
    """The bytes of a consensus FASTA file (utils.write_fasta_record): what snp_matrix copies into snpma.fasta."""
    n = len(seq)
    head = (">%s\n" % name).encode("utf-8")
    if n == 0:
        return head
    full, tail = divmod(n, 60)
    body = np.full((full + (1 if tail else 0), 61), 0x0A, dtype=np.uint8)
    if full:
        body[:full, :60] = seq[:full * 60].reshape(full, 60)
    out = body[:full].tobytes()
    if tail:
        out += seq[full * 60:].tobytes() + b"\n"
    return head + out


def _call_scattered(dev, ss, prm, res_idx, ptrs, sizes, d_base, d_filt, d_status, d_counts, d_line, S, want_vcf, torch, want_depth_sum=False):
    """A group in which only some samples are resident: they are called into temporary arrays and scattered to their rows."""
    m = len(res_idx)
    tb = torch.empty((m, S), dtype=torch.uint8, device="cuda")
    tf = torch.empty((m, S), dtype=torch.uint8, device="cuda")
    tl = torch.zeros((m, S), dtype=torch.int64, device="cuda")
    ts = torch.empty((m, 4), dtype=torch.int64, device="cuda")
    tc = torch.empty((m, S, 128), dtype=torch.uint8, device="cuda") if want_vcf else None
    dev.call_consensus_many_dev(ss, ptrs, sizes, prm, tb.data_ptr(), tf.data_ptr(), ts.data_ptr(), d_counts=tc.data_ptr() if want_vcf else 0,
                                d_line_off=tl.data_ptr(), want_depth_sum=want_depth_sum)
    idx = torch.tensor(res_idx, dtype=torch.int64, device="cuda")
    d_base[idx, :S] = tb
    d_filt[idx, :S] = tf
    d_line[idx, :S] = tl
    d_status[idx] = ts
    if want_vcf:
        d_counts[idx, :S] = tc


def add_arguments(sub):
    sub.add_argument(dest="sampleDirsFile", type=str, help="Relative or absolute path to file containing a list of directories -- one per sample")
    sub.add_argument(dest="referenceFile", type=str, help="Relative or absolute path to the reference fasta file")
    sub.add_argument("-f", "--force", dest="forceFlag", action="store_true", help="Force processing even when result files already exist and are newer than inputs")
    sub.add_argument("--workDir", dest="workDir", type=str, default=None, metavar="DIR", help="Directory of the top-level output files (default: the directory of sampleDirsFile)")
    sub.add_argument("--pileupName", dest="pileupName", type=str, default="reads.all.pileup", metavar="NAME", help="File name of the genome-wide pileup file in each sample directory.")
    for name, env in (("filterRegionsExtraParams", "FilterRegions_ExtraParams"), ("mergeSitesExtraParams", "MergeSites_ExtraParams"),
                      ("callConsensusExtraParams", "CallConsensus_ExtraParams"), ("varscanExtraParams", "VarscanMpileup2snp_ExtraParams")):
        sub.add_argument("--" + name, dest=name, type=str, default=None, metavar="STRING",
                         help="Options of that step, as the configuration file gives them (default: the environment variable %s)" % env)
    sub.add_argument("--noConsensusVcf", dest="noConsensusVcf", action="store_true", help="Do not write consensus.vcf / consensus_preserved.vcf")
    sub.add_argument("--residentBytes", dest="residentBytes", type=int, default=0, metavar="INT", help="Device memory for resident pileups (0 = what is free, less 24 GiB); files past it are streamed twice")
    sub.add_argument("--groupBytes", dest="groupBytes", type=int, default=0, metavar="INT", help="Host bytes of per-site results per group of samples (default 1.5 GiB)")
    sub.add_argument("--writerThreads", dest="writerThreads", type=int, default=0, metavar="INT", help="Host threads that write the consensus files (0 = up to 64)")

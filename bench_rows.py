"""bench_rows.py — the side rows of bench.py: everything that is measured AFTER the timed region of the headline step and goes
into the detail file (gpurun_out/bench_detail.json), a few numbers of which the compact line quotes.

  pipeline_from_files   the rank's shard as ONE job from files to files (`cfsan_snp_pipeline hot_path_batch`), the separate
                        subcommands on the same tree beside it;
  secondary             pairwise SNP distances/s of the distance step alone at BASELINE configs[4] shape (10 000 x 200 000);
  aux_steps             K3 / K4 (site union, dense windows, region merge, in-region test) at configs[3] scale;
  scan_shapes           the pileup-scan kernel on its weak shapes (shallow / deep pileups, CR LF, many contigs);
  end_to_end            pileup FILES in the page cache -> consensus bytes on the host through the streamed ingestion;
  call_variants         the call paths the headline leaves out: per-site counts, the strict caller, --vcfAllPos;
  site_calling          phase-1 site calling;
  cpu_baseline          the CPU oracle on samples of the same batch: 1 core, one process per sample, the distance loop.

oracle/ is imported here only by the checker legs (outside every timed region) and by cpu_baseline.
"""
import json
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
PROFILE_ROUNDS = ("r6", "r5", "r3", "r2", "r1")   # where a committed PMC pass of the same workload may be found, newest first


def _oracle_worker(job):
    """One call_consensus of the CPU oracle in its own process (bench cpu_baseline.parallel)."""
    path, positions = job
    from oracle import pileup_oracle as po
    with open(path, "rb") as f:
        data = f.read()
    cons, _ = po.call_consensus_sites(data, [(b"synth_chr1", p) for p in positions], set(), po.CallerParams(0, 0.6, 3, 0, 0.0))
    return cons


def effective_cores():
    """CPUs this process may really use: the scheduler's affinity mask, capped by the cgroup's CPU quota (a container that sees
    256 CPUs may be allowed the time of 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                words = f.read().split()
            if path.endswith("cpu.max"):
                if words and words[0] != "max":
                    n = min(n, max(1, int(int(words[0]) / int(words[1]))))
            else:
                quota = int(words[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                    period = int(f2.read().split()[0])
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _scratch_dir(need_bytes):
    """A directory on a regular file system with room for the pileup files (page cache), else tmpfs.  The FIRST read of
    freshly written tmpfs pages is serialised in the kernel (~15 GB/s on the bench box whatever the thread count; later
    passes, and every pass over ordinary page-cache files, are not), so tmpfs comes second."""
    import shutil
    import tempfile
    for cand in (tempfile.gettempdir(), "/dev/shm"):
        try:
            if shutil.disk_usage(cand).free > 2 * need_bytes:
                return cand
        except OSError:
            pass
    return None


def end_to_end(d, ss, prm, pile, offs, sizes, bases, n_files, S):
    """Page-cache files -> consensus bytes: the rate a run over more samples than fit in HBM proceeds at.  The files are
    written first, one warm-up file goes through (pinned staging allocation), then all of them are timed in one
    snpgpu_call_consensus_files call and compared with the resident results; the yardstick is a pinned host-to-device
    copy measured in the same process."""
    import shutil
    import tempfile
    import torch
    need = int(sum(sizes[:n_files])) + (64 << 20)
    base_dir = _scratch_dir(need)
    if base_dir is None:
        return {"skipped": "no room for %d bytes of pileup files" % need}
    tmpdir = tempfile.mkdtemp(prefix="snpbench_e2e_", dir=base_dir)
    try:
        paths = []
        for i in range(n_files):
            path = os.path.join(tmpdir, "s%d.pileup" % i)
            with open(path, "wb") as f:
                f.write(pile[int(offs[i]):int(offs[i]) + sizes[i]].cpu().numpy().tobytes())
            paths.append(path)
        # pinned host -> device copy rate (the ceiling of this path): the best of several shapes of the copy, see pinned_h2d_gbps
        h2d = pinned_h2d_gbps(torch)
        d.call_consensus_files(ss, paths[:1], prm)                                  # warm-up: pinned staging, device slots
        # two passes over the same files, the better one reported (both listed): single passes spread between 42 and 56 GB/s
        # on the bench box whatever the reader count (>= 8) and whichever socket wrote the files (tools/e2e_readers.py)
        passes = []
        st = None
        for _ in range(2):
            res, rcs, st_i = d.call_consensus_files(ss, paths, prm)
            ok = all(int(rc) == 0 for rc in rcs) and all(bytes(res[i].bases) == bytes(bases[i].cpu().numpy()) for i in range(n_files))
            if not ok:
                raise SystemExit("streamed consensus differs from the resident one")
            passes.append(st_i.bytes / st_i.seconds / 1e9)
            if st is None or st_i.seconds < st.seconds:
                st = st_i
        gbps = st.bytes / st.seconds / 1e9
        # the preserved flow's shape (run.py:712-718: call_consensus -e var.flt_removed.vcf): every file with its OWN exclude list
        # — 1 500 slots each, as a sample's removed positions — in the same single call; the excluded positions that have a pileup
        # line must come back as '-' with the Region bit, everything else as in the plain pass
        rng = np.random.default_rng(11)
        excl = [np.sort(rng.choice(S, size=min(1500, S), replace=False)) for _ in range(n_files)]
        res_e, rcs_e, st_e = d.call_consensus_files(ss, paths, prm, exclude=excl)
        passes_e = [st_e.bytes / st_e.seconds / 1e9]
        res_e2, rcs_e2, st_e2 = d.call_consensus_files(ss, paths, prm, exclude=excl)          # two passes, as for the plain call
        passes_e.append(st_e2.bytes / st_e2.seconds / 1e9)
        if st_e2.seconds < st_e.seconds:
            res_e, rcs_e, st_e = res_e2, rcs_e2, st_e2
        region_bit = 0x20
        for i in range(n_files):
            plain, got = res[i], res_e[i]
            mask = np.zeros(S, dtype=bool)
            mask[excl[i]] = True
            has_line = np.asarray(plain.bases) != 0x2D                           # (a '-' of the plain pass stays '-')
            want = np.where(mask, 0x2D, np.asarray(plain.bases)).astype(np.uint8)
            if int(rcs_e[i]) != 0 or not np.array_equal(np.asarray(got.bases), want) or \
                    not ((np.asarray(got.filters)[mask & has_line] & region_bit) != 0).all() or \
                    not np.array_equal(np.asarray(got.filters)[~mask], np.asarray(plain.filters)[~mask]):
                raise SystemExit("the pass with per-file exclude lists differs from the plain pass outside the excluded positions")
        gbps_e = st_e.bytes / st_e.seconds / 1e9
        return {
            "what": "%d pileup files in the page cache (%s) -> consensus bytes on the host, one snpgpu_call_consensus_files call"
                    % (n_files, base_dir),
            "with_per_file_exclude_lists": {"pileup_gb_per_sec": gbps_e, "seconds": st_e.seconds, "over_plain_pass": gbps_e / gbps,
                                            "excluded_positions_per_file": int(len(excl[0])), "passes_gb_per_sec": passes_e, "checked": True},
            "files": n_files, "bytes": int(st.bytes), "seconds": st.seconds, "pileup_gb_per_sec": gbps,
            "consensus_bases_per_sec": n_files * S / st.seconds, "samples_per_sec": n_files / st.seconds,
            "pinned_h2d_gb_per_sec": h2d, "frac_of_pinned_h2d": over_link(gbps, h2d), "pinned_h2d_probe_is_a_ceiling_here": over_link(gbps, h2d) is not None,
            "pinned_h2d_probe": pinned_h2d_probe(), "passes_gb_per_sec": passes,
            "chunk_bytes": int(st.chunk_bytes), "reader_threads": int(st.n_readers), "staging_buffers": int(st.n_staging),
            "seconds_waiting_for_readers": st.seconds_waiting_for_readers,
            "seconds_waiting_for_device": st.seconds_waiting_for_device, "matches_resident": True,
        }
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)


def site_calling(d, pile, offs, sizes, n_files):
    """Phase-1 site calling on pileup FILES in the page cache: file -> var.flt.vcf through
    varscan.mpileup2snp (reader threads + copy, k_varscan_scan + k_varscan_finish, host finish), best of two passes; every
    record of the first file is checked against the CPU restatement (oracle/varscan_oracle.py) on the record's own line,
    and the restatement is timed on the first lines of that file for the CPU column."""
    import shutil
    import tempfile
    from oracle import varscan_oracle as vo
    from snp_pipeline_amd import varscan
    need = int(sum(sizes[:n_files])) + (64 << 20)
    base_dir = _scratch_dir(need)
    if base_dir is None:
        return {"skipped": "no room for %d bytes of pileup files" % need}
    tmpdir = tempfile.mkdtemp(prefix="snpbench_sites_", dir=base_dir)
    try:
        paths = []
        first = None
        for i in range(n_files):
            path = os.path.join(tmpdir, "s%d.pileup" % i)
            data = pile[int(offs[i]):int(offs[i]) + sizes[i]].cpu().numpy().tobytes()
            if i == 0:
                first = data
            with open(path, "wb") as f:
                f.write(data)
            paths.append(path)
        extra = "--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5"                 # snppipeline.conf:199
        opts = varscan.Options(extra)
        vcf = os.path.join(tmpdir, "var.flt.vcf")
        vcfs = [os.path.join(tmpdir, "s%d.var.flt.vcf" % i) for i in range(n_files)]
        varscan.mpileup2snp_files(d, paths[:2], vcfs[:2], opts)                        # warm-up: pinned staging, both device slots
        best, rows, lines, passes = None, 0, 0, []
        for _ in range(2):
            t0 = time.perf_counter()
            res = varscan.mpileup2snp_files(d, paths, vcfs, opts)                      # one streamed call for all files
            dt = time.perf_counter() - t0
            lines, rows = sum(r[0] for r in res), sum(r[1] for r in res)
            passes.append(dt)
            best = dt if best is None or dt < best else best
        # parity spot check: the first file's rows against the restatement, line by line
        varscan.mpileup2snp(d, paths[0], vcf, opts)
        got = [ln for ln in open(vcf).read().splitlines(True) if not ln.startswith("#")]
        recs, _ = d.varscan_file(paths[0], opts.device_params())
        prm = vo.Params(**vo.PIPELINE_DEFAULTS)
        ok = len(recs) == len(got)
        for k in range(len(got)):
            off = int(recs["line_off"][k])
            f = first[off:first.index(b"\n", off)].split(b"\t")
            r = vo.call_line(f[2].decode(), int(f[3]), f[4], f[5], prm)
            ok = ok and r is not None and vo.vcf_row(f[0].decode(), f[1].decode(), r) == got[k]
        if not ok:
            raise SystemExit("site calling differs from its CPU restatement")
        # the CPU restatement on the first ~8 MB of the file (whole lines), one core
        cut = first.rfind(b"\n", 0, 8 << 20) + 1
        t0 = time.perf_counter()
        vo.mpileup2snp(first[:cut], prm)
        cpu_s = time.perf_counter() - t0
        nbytes = int(sum(sizes[:n_files]))
        # the kernels alone, over ALL resident samples of the shard in ONE launch (snpgpu_varscan_batch_dev: k_varscan_scan — one pass over
        # the text, every wave walks its own candidate lines — and k_varscan_finish), HIP events on the launch stream around them.
        # Algorithmic bytes = the text, once.  Beside it: the same for one sample per launch (a 0.15 ms grid pays its ramp and tail).
        dprm = opts.device_params()
        n_res = len(sizes)
        ptrs = [pile.data_ptr() + int(offs[i]) for i in range(n_res)]
        lens = [int(sizes[i]) for i in range(n_res)]
        res_b = d.varscan_batch_dev(ptrs, lens, dprm, capacity=8192)                   # warm-up, and the answer of sample 0 against the file route's
        if isinstance(res_b[0], Exception) or res_b[0][0].tobytes() != recs.tobytes():
            raise SystemExit("site calling over resident samples differs from the file route")
        d.kernel_timing(True)
        d.kernel_time_ms(3)
        reps_b = 5
        for _ in range(reps_b):
            d.varscan_batch_dev(ptrs, lens, dprm, capacity=8192)
        kb_ms, kb_n = d.kernel_time_ms(3)                       # (a call makes one launch per dozen 30x samples: kb_n launches in all)
        kb_avg = kb_ms / reps_b                                 # per call over the whole shard
        tot_b = int(sum(lens))
        k_gbs = tot_b / (kb_avg * 1e-3) / 1e9 if kb_avg > 0 else 0.0
        k_avg = kb_avg / max(n_res, 1)
        k_n = kb_n
        # a dozen samples per call (one launch each), the host's read-back between the calls: what the launches of the long call settle at
        # once the clocks have (a VALU-bound kernel is clocked down a few milliseconds into a sustained run that follows an idle stretch)
        n12 = min(12, n_res)
        d.kernel_time_ms(3)
        for _ in range(6):
            d.varscan_batch_dev(ptrs[:n12], lens[:n12], dprm, capacity=8192)
        k12_ms, _k12_n = d.kernel_time_ms(3)
        k12_avg = k12_ms / 6.0 / max(n12, 1)
        k12_gbs = (sum(lens[:n12]) / n12) / (k12_avg * 1e-3) / 1e9 if k12_avg > 0 else 0.0
        d.varscan_dev(ptrs[0], lens[0], dprm)
        d.kernel_time_ms(3)
        for _ in range(10):
            d.varscan_dev(ptrs[0], lens[0], dprm)
        k1_ms, k1_n = d.kernel_time_ms(3)
        d.kernel_timing(False)
        k1_avg = k1_ms / max(k1_n, 1)
        k1_gbs = lens[0] / (k1_avg * 1e-3) / 1e9 if k1_avg > 0 else 0.0
        # the same shape (5 Mbp x 30x): sample 0 of the profile's generator
        vs_traffic, vs_src = committed_traffic("varscan_traffic.json", lambda vt: abs(vt.get("bytes", 0) - int(sizes[0])) <= int(sizes[0]) // 100)
        return {
            # headline = the shape the pipeline runs: snpgpu_pileups_ingest / snpgpu_varscan_files launch the kernels once per FILE as it
            # arrives (hot_path.py stage 1, call_sites_batch); snpgpu_varscan_batch_dev (many resident samples per launch) is an entry
            # point of the ABI that no subcommand calls — its figure is listed beside it and labelled so
            "roofline": {"kernels": "k_varscan_scan + k_varscan_finish, one resident sample per launch (what the ingest of hot_path_batch and call_sites_batch do)",
                         "bound": "hbm", "achieved": k1_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k1_gbs / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_file": int(lens[0]), "avg_ms_per_file": k1_avg, "files_timed": int(k1_n),
                         "batch_over_the_shard_not_used_by_the_pipeline": {
                             "entry_point": "snpgpu_varscan_batch_dev: one call over the shard's %d resident samples, a launch per dozen of them" % n_res,
                             "achieved": k_gbs, "frac": k_gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_call": tot_b, "samples_per_call": n_res,
                             "avg_ms_per_call": kb_avg, "calls_timed": reps_b, "launches_per_call": int(kb_n) // reps_b, "avg_ms_per_file": k_avg},
                         "twelve_samples_per_call_not_used_by_the_pipeline": {"achieved": k12_gbs, "frac": k12_gbs / HBM_PEAK_GBS, "avg_ms_per_file": k12_avg, "calls_timed": 6},
                         "traffic": (vs_traffic or {}).get("traffic_bytes_per_file"), "traffic_measured_on_bytes": (vs_traffic or {}).get("bytes"),
                         "traffic_over_algorithmic": (vs_traffic or {}).get("traffic_over_algorithmic"),
                         "traffic_source": ("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)" % vs_src) if vs_traffic else None,
                         "note": "resident samples, HIP events around the launches on their stream; the files -> var.flt.vcf rate below is bound by the host link"},
            "what": "%d pileup files in the page cache -> var.flt.vcf each (VarScan mpileup2snp's job, %s), one snpgpu_varscan_files call" % (n_files, extra),
            "files": n_files, "bytes": nbytes, "seconds": best, "pileup_gb_per_sec": nbytes / best / 1e9, "samples_per_sec": n_files / best,
            "pileup_lines_per_sec": lines / best, "sites_written": rows, "rows_equal_cpu_restatement": True,
            "passes_gb_per_sec": [nbytes / x / 1e9 for x in passes],
            "cpu_port": {"pileup_gb_per_sec": cut / cpu_s / 1e9, "cores": 1, "sample": "the first %d bytes of one file" % cut,
                         "kind": "port (oracle/varscan_oracle.py; the reference runs the VarScan jar here, which this image lacks)"},
        }
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)


# ---- pipeline_from_files: the shard as one job from files to files ------------------------------------------------------------
FILTER_EXTRA = "--edge_length 500 --window_size 1000 125 15 --max_snp 3 2 1 --mode all"       # snppipeline.conf:211
CONSENSUS_EXTRA = "--minConsFreq 0.6 --minConsDpth 3"                                          # snppipeline.conf:249
VARSCAN_EXTRA = "--min-avg-qual 15 --min-var-freq 0.90 --min-reads2 5"                        # snppipeline.conf:199
TOP_LEVEL_FILES = ("snplist.txt", "snplist_preserved.txt", "snpma.fasta", "snpma_preserved.fasta", "snp_distance_pairwise.tsv",
                   "snp_distance_matrix.tsv", "snp_distance_pairwise_preserved.tsv", "snp_distance_matrix_preserved.tsv",
                   "referenceSNP.fasta", "referenceSNP_preserved.fasta")
PER_SAMPLE_FILES = ("var.flt.vcf", "var.flt_preserved.vcf", "var.flt_removed.vcf", "consensus.fasta", "consensus.vcf",
                    "consensus_preserved.fasta", "consensus_preserved.vcf")


def call_variants(d, ss, prm, pile, offs, sizes, S, dev, torch, pos):
    """The call paths the headline does not time, on the same resident shard, HIP events of the context around the
    scan and the call kernels:
      call_with_counts  per-site counts for consensus.vcf — the reference's default configuration writes it (snppipeline.conf:249,
                        run.py:709): k_call_lanes<..., true> + k_call_sites, one 128-byte record per (sample, site);
      strict            the strict caller set of SURVEY 8(d): -q 15 -c 0.9 -D 5 -d 2 -b 0.1, no counts;
      all_positions     --vcfAllPos (call_consensus.py:148, pileup.py:418-421): a Record from EVERY line of one sample (5 M), from
                        its file in the page cache through snpgpu_call_all_lines_file (line index + k_call_sites + the records back).
    K2 roofline: algorithmic bytes = the bytes of the lines that are looked at (matched lines x the batch's mean line length; every
    line of the file for all_positions) + the records / bytes written."""
    import shutil
    import tempfile
    from oracle import pileup_oracle as po
    B = len(sizes)
    ptrs = [pile.data_ptr() + int(offs[i]) for i in range(B)]
    lens = [int(x) for x in sizes]
    pile_bytes = int(sum(lens))
    out = {}
    bases = torch.empty((B, S), dtype=torch.uint8, device="cuda")
    filt = torch.empty((B, S), dtype=torch.uint8, device="cuda")
    status = torch.empty((B, 4), dtype=torch.int64, device="cuda")

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        d.kernel_timing(True)
        d.kernel_time_ms(0), d.kernel_time_ms(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        scan_ms, scan_n = d.kernel_time_ms(0)
        call_ms, call_n = d.kernel_time_ms(1)
        d.kernel_timing(False)
        return a.elapsed_time(b) / reps, scan_ms / max(scan_n, 1), call_ms / max(call_n, 1)

    def row(ms, scan_ms, call_ms, positions, matched, looked_bytes, written_bytes, what):
        algo = looked_bytes + written_bytes
        gbs = algo / (call_ms * 1e-3) / 1e9 if call_ms > 0 else 0.0
        return {"what": what, "ms_per_step": ms, "positions_per_sec": positions / (ms * 1e-3), "positions": positions, "matched_lines": matched,
                "k_scan_wave_ms": scan_ms, "call_kernels_ms": call_ms,
                "roofline": {"kernels": "K2: k_call_lanes x3 + k_call_sites (everything between the scan and the results)", "bound": "hbm", "achieved": gbs,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes": algo,
                             "of_which_line_bytes": looked_bytes, "of_which_written": written_bytes, "traffic": None}}

    # ---- per-site counts --------------------------------------------------------------------------------------------
    counts = torch.empty((B, S, dev.COUNTS_DTYPE.itemsize), dtype=torch.uint8, device="cuda")

    def with_counts():
        d.call_consensus_many_dev(ss, ptrs, lens, prm, bases.data_ptr(), filt.data_ptr(), status.data_ptr(), d_counts=counts.data_ptr())
    ms, scan_ms, call_ms = timed(with_counts, 3)
    st = status.cpu().numpy()
    n_lines, matched = int(st[:, 1].sum()), int(st[:, 2].sum())
    mean_line = pile_bytes / max(n_lines, 1)
    # check: the records of sample 0 at 200 sites against the oracle on their own lines
    rec = np.frombuffer(counts[0].cpu().numpy().tobytes(), dtype=dev.COUNTS_DTYPE)
    text = pile[int(offs[0]):int(offs[0]) + lens[0]].cpu().numpy().tobytes()
    p0 = po.CallerParams(0, 0.6, 3, 0, 0.0)
    cut = text.rfind(b"\n", 0, 24 << 20) + 1                  # the oracle on the sample's first 24 MB (whole lines), and the sites that lie in them
    last_pos = int(text[text.rfind(b"\n", 0, cut - 1) + 1:cut].split(b"\t")[1])
    inside = pos[pos < last_pos]
    sub = [(b"synth_chr1", int(p)) for p in inside[::max(len(inside) // 200, 1)]]
    _, detail = po.call_consensus_sites(text[:cut], sub, set(), p0)
    slot = {k: i for i, k in enumerate(ss.key_tuples())}
    for key, (r, base, mask) in detail.items():
        c = rec[slot[key]]
        if (int(c["raw_depth"]), int(c["good_depth"]), int(c["fwd_good_depth"]), int(c["rev_good_depth"]), int(c["cons_base"]), int(c["filters"])) != \
                (r.raw_depth, r.good_depth, r.forward_good_depth, r.reverse_good_depth, base, mask):
            raise SystemExit("per-site counts differ from the oracle at %r" % (key,))
    out["call_with_counts"] = row(ms, scan_ms, call_ms, B * S, matched, int(matched * mean_line), matched * dev.COUNTS_DTYPE.itemsize + 2 * B * S,
                                  "%d samples x %d sites with per-site count records (consensus.vcf's input): scan + call, one group per launch" % (B, S))
    out["call_with_counts"]["records_checked_against_oracle"] = len(detail)
    del counts, rec

    # ---- the strict caller ------------------------------------------------------------------------------------------------
    strict = dev.make_params(15, 0.9, 5, 2, 0.1)
    sizes_np = np.asarray(lens, dtype=np.uint64)

    def strict_call():
        d.call_consensus_batch_dev(ss, pile.data_ptr(), offs[:B], strict, bases.data_ptr(), filt.data_ptr(), status.data_ptr(), sizes=sizes_np)
    ms, scan_ms, call_ms = timed(strict_call, 5)
    ps = po.CallerParams(15, 0.9, 5, 2, 0.1)
    want, _ = po.call_consensus_sites(text[:cut], sub, set(), ps)
    got = bytes(int(x) for x in bases[0].cpu().numpy()[[slot[k] for k in sub]])
    if got != want:
        raise SystemExit("the strict caller differs from the oracle")
    out["strict"] = row(ms, scan_ms, call_ms, B * S, matched, int(matched * mean_line), 2 * B * S,
                        "%d samples x %d sites, caller -q 15 -c 0.9 -D 5 -d 2 -b 0.1 (SURVEY 8d), no count records" % (B, S))
    out["strict"]["sites_checked_against_oracle"] = len(sub)
    # HBM traffic of the call kernels from the committed PMC passes of the same shape (rocprofv3 cannot run inside this process)
    for key in ("call_with_counts", "strict"):
        r = out[key]["roofline"]
        ct, src = committed_traffic("call_traffic.json", lambda c: (c.get(key) or {}).get("algorithmic_bytes") and
                                    abs(c[key]["algorithmic_bytes"] - r["algorithmic_bytes"]) <= r["algorithmic_bytes"] // 50)
        if ct:
            r["traffic"] = ct[key]["traffic_bytes_per_step"]
            r["traffic_over_algorithmic"] = ct[key]["traffic_over_algorithmic"]
            r["traffic_source"] = "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % src

    # ---- every line of one sample ---------------------------------------------------------------------------------------------
    base_dir = _scratch_dir(lens[0] + (64 << 20))
    if base_dir is None:
        out["all_positions"] = {"skipped": "no room for one pileup file"}
        return out
    tmpdir = tempfile.mkdtemp(prefix="snpbench_allpos_", dir=base_dir)
    try:
        path = os.path.join(tmpdir, "reads.all.pileup")
        with open(path, "wb") as f:
            f.write(text)
        lines0 = int(st[0, 1])
        # records back: the 32-byte line records (snpgpu_call_all_lines_compact_file), which is what crosses the host link since round 6; the
        # 128-byte form (snpgpu_call_all_lines_file, round 5's row) timed beside it
        d.call_all_lines_compact(ss, path, prm, capacity=lines0)                  # warm-up: scratch, pinned staging
        d.kernel_timing(True)
        d.kernel_time_ms(1)
        t0 = time.perf_counter()
        off, lrec, widx, wide = d.call_all_lines_compact(ss, path, prm, capacity=lines0, wide_capacity=max(lines0 // 16, 4096))
        wall = time.perf_counter() - t0
        call_ms, call_n = d.kernel_time_ms(1)
        d.kernel_timing(False)
        call_ms = call_ms / max(call_n, 1)
        _, recs = dev.expand_line_records(lrec, widx, wide)
        t0 = time.perf_counter()
        off_full, _, recs_full = d.call_all_lines(ss, path, prm, capacity=lines0, check=False)
        wall_full = time.perf_counter() - t0
        recs_full["n_symbols"] &= 0xFF                                           # (the index of a spill record differs from call to call; the synthetic lines have none)
        recs["n_symbols"] &= 0xFF
        if not np.array_equal(off, off_full) or recs.tobytes() != recs_full.tobytes():
            raise SystemExit("--vcfAllPos: the 32-byte line records differ from the full ones")
        del recs_full, off_full
        # check: 300 lines spread over the file, each against the oracle's Record of that very line
        for k in range(0, len(off), max(len(off) // 300, 1)):
            o = int(off[k]) - 1
            ln = text[o:text.index(b"\n", o)]
            r = po.parse_record(po.split_fields(ln), 0)
            if (int(recs[k]["raw_depth"]), int(recs[k]["good_depth"]), int(recs[k]["fwd_good_depth"])) != (r.raw_depth, r.good_depth, r.forward_good_depth):
                raise SystemExit("--vcfAllPos record %d differs from the oracle" % k)
        # file to file: reads.all.pileup -> consensus.vcf with a row per line (what `call_consensus --vcfAllPos` does with the records)
        import argparse
        from snp_pipeline_amd import vcf_writer
        cc = argparse.Namespace(minBaseQual=0, minConsFreq=0.6, minConsDpth=3, minConsStrdDpth=0, minConsStrdBias=0.0, vcfRefName="ref.fasta",
                                vcfPreserveRefCase=False, vcfFailedSnpGt=".")
        vcf = os.path.join(tmpdir, "consensus.vcf")
        vcf_writer.write_all_positions_vcf_from_pileup(d, ss, vcf, "s0", cc, path, prm)                  # warm-up: page cache of the output
        t0 = time.perf_counter()
        n_l, n_rows = vcf_writer.write_all_positions_vcf_from_pileup(d, ss, vcf, "s0", cc, path, prm)
        wall_vcf = time.perf_counter() - t0
        vcf_bytes = os.path.getsize(vcf)
        # its rows against the row-by-row writer: the ends of the file and a sample spread over all of it
        filt = [n for n, _ in vcf_writer.filter_descriptions(0.6, 3, 0, 0.0)]
        with open(vcf, "rb") as f:
            rows = [ln for ln in f.read().split(b"\n") if ln and not ln.startswith(b"#")]
        if n_rows != len(off) or len(rows) != len(off):
            raise SystemExit("--vcfAllPos: %d rows for %d lines" % (len(rows), len(off)))
        stride = max(len(off) // 4000, 1)                                        # the first and the last 1 000 rows and 4 000 spread over every piece of the read-back
        for k in sorted(set(list(range(0, min(1000, len(off)))) + list(range(0, len(off), stride)) + list(range(max(0, len(off) - 1000), len(off))))):
            o = int(off[k]) - 1
            f0, f1 = text[o:o + 256].split(None, 2)[:2]
            if rows[k].decode() != vcf_writer.row_from_counts(f0.decode(), int(f1), recs[k], filt, False, ".", spill=d.last_spill):
                raise SystemExit("--vcfAllPos row %d differs from the row-by-row writer" % k)
        written = len(off) * (dev.LINE_DTYPE.itemsize + 8) + len(widx) * (dev.COUNTS_DTYPE.itemsize + 4)
        gbs = (lens[0] + len(off) * (dev.COUNTS_DTYPE.itemsize + 8 + 1)) / (call_ms * 1e-3) / 1e9 if call_ms > 0 else 0.0
        out["all_positions"] = {
            "what": "one sample, a record from every one of its %d lines (--vcfAllPos): file in the page cache -> 32-byte records on the host" % len(off),
            "ms_per_step": wall * 1e3, "positions_per_sec": len(off) / wall, "positions": int(len(off)), "call_kernel_ms": call_ms,
            "bytes_back_over_the_host_link": written, "wide_lines": int(len(widx)),
            "with_128_byte_records_ms": wall_full * 1e3, "with_128_byte_records_bytes_back": len(off) * (dev.COUNTS_DTYPE.itemsize + 8 + 1),
            "file_to_vcf_file": {"ms": wall_vcf * 1e3, "rows": int(n_rows), "vcf_bytes": vcf_bytes, "rows_per_sec": n_rows / wall_vcf,
                                 "rows_checked_against_the_row_by_row_writer": min(6000, len(off)),
                                 "what": "reads.all.pileup (page cache) -> consensus.vcf with a row per line, snpgpu_write_all_positions_vcf: records back in pieces, "
                                         "rows formatted and written by the library's host threads"},
            "roofline": {"kernels": "K2 over a line list: k_call_lanes x3 + k_call_sites", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes": lens[0] + len(off) * (dev.COUNTS_DTYPE.itemsize + 8 + 1), "of_which_line_bytes": lens[0],
                         "traffic": None},
            "note": "the wall time holds the file read, its copy to the device, the line index, the call, the packing and %.0f MB of records back over the host link" % (written / 1e6)}
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)
    return out


def run_cli(line, verbose=0):
    """One subcommand in this process (\\x00 stands for a blank inside an argument).  Returns its wall time."""
    from snp_pipeline_amd import cfsan_snp_pipeline as cli
    a = cli.parse_argument_list([w.replace("\x00", " ") for w in line.split()])
    a.verbose = verbose
    t = time.perf_counter()
    cli.run_command_from_args(a)
    return time.perf_counter() - t


def hot_path_line(dirs_file, ref_path, extra=""):
    q = lambda x: x.replace(" ", "\x00")     # noqa: E731
    return ("hot_path_batch -f %s %s --filterRegionsExtraParams=%s --callConsensusExtraParams=%s --varscanExtraParams=%s%s"
            % (dirs_file, ref_path, q(FILTER_EXTRA), q(CONSENSUS_EXTRA), q(VARSCAN_EXTRA), extra))


def separate_steps(work, ref_path, dirs_file):
    """The same files through the separate subcommands (their batch forms: the per-sample CLI of run.py:704-718 adds a process
    start per sample on top): call_sites_batch, filter_regions, merge_sites x 2, call_consensus_batch x 2, snp_matrix x 2,
    snp_reference x 2, distance x 2.  Returns {step: seconds}."""
    os.environ["VarscanMpileup2snp_ExtraParams"] = VARSCAN_EXTRA
    t = {}
    t["call_sites_batch"] = run_cli("call_sites_batch %s %s" % (ref_path, dirs_file))       # (no -f: that would also re-run samtools)
    t["filter_regions"] = run_cli("filter_regions -f -n var.flt.vcf %s %s %s" % (dirs_file, ref_path, FILTER_EXTRA))
    t["merge_sites"] = run_cli("merge_sites -f -n var.flt.vcf -o %s/snplist.txt %s %s.OrigVCF.filtered" % (work, dirs_file, dirs_file))
    t["merge_sites_preserved"] = run_cli("merge_sites -f -n var.flt_preserved.vcf -o %s/snplist_preserved.txt %s %s.PresVCF.filtered" % (work, dirs_file, dirs_file))
    t["call_consensus_batch"] = run_cli("call_consensus_batch -f -l %s/snplist.txt -o consensus.fasta --vcfRefName ref.fasta %s --vcfFileName consensus.vcf %s"
                                        % (work, CONSENSUS_EXTRA, dirs_file))
    t["call_consensus_batch_preserved"] = run_cli("call_consensus_batch -f -l %s/snplist_preserved.txt -o consensus_preserved.fasta -e var.flt_removed.vcf "
                                                  "--vcfRefName ref.fasta %s --vcfFileName consensus_preserved.vcf %s" % (work, CONSENSUS_EXTRA, dirs_file))
    for sfx, flt in (("", "OrigVCF"), ("_preserved", "PresVCF")):
        t["snp_matrix" + sfx] = run_cli("snp_matrix -f -c consensus%s.fasta -o %s/snpma%s.fasta %s.%s.filtered" % (sfx, work, sfx, dirs_file, flt))
        t["snp_reference" + sfx] = run_cli("snp_reference -f -l %s/snplist%s.txt -o %s/referenceSNP%s.fasta %s" % (work, sfx, work, sfx, ref_path))
        t["distance" + sfx] = run_cli("distance -f -p %s/snp_distance_pairwise%s.tsv -m %s/snp_distance_matrix%s.tsv %s/snpma%s.fasta" % (work, sfx, work, sfx, work, sfx))
    return t


def output_digests(work, dirs):
    import hashlib
    h = {}
    for name in TOP_LEVEL_FILES:
        with open(os.path.join(work, name), "rb") as f:
            h[name] = hashlib.sha256(f.read()).hexdigest()
    for name in PER_SAMPLE_FILES:
        m = hashlib.sha256()
        for sdir in dirs:
            with open(os.path.join(sdir, name), "rb") as f:
                m.update(f.read())
        h["samples/*/" + name] = m.hexdigest()
    return h


def write_sample_tree(base_dir, refh, G, sample_bytes, n, contig="synth_chr1"):
    """reference/ref.fasta + samples/sNNNN/reads.all.pileup for n samples; sample_bytes(i) -> the pileup of sample i as a host
    array.  Returns (tmpdir, ref path, dirs file, sample dirs, total pileup bytes)."""
    import concurrent.futures
    import tempfile
    tmpdir = tempfile.mkdtemp(prefix="snpbench_pipeline_", dir=base_dir)
    os.makedirs(os.path.join(tmpdir, "reference"))
    ref_path = os.path.join(tmpdir, "reference", "ref.fasta")
    seq = refh[1:G + 1].tobytes().decode()
    with open(ref_path, "w") as f:
        f.write(">%s\n" % contig)
        f.write("\n".join(seq[i:i + 60] for i in range(0, G, 60)) + "\n")
    old = time.time() - 1000
    os.utime(ref_path, (old, old))

    def write(path, arr):
        with open(path, "wb") as f:
            f.write(memoryview(arr))

    dirs, total, futures = [], 0, []
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as pool:
        for i in range(n):
            host = sample_bytes(i)
            sdir = os.path.join(tmpdir, "samples", "s%04d" % i)
            os.makedirs(sdir)
            bam = os.path.join(sdir, "reads.sorted.deduped.indelrealigned.bam")
            with open(bam, "wb") as f:
                f.write(b"placeholder: the pileup is newer, samtools is not run (call_sites.py:70-72)")
            os.utime(bam, (old, old))
            futures.append(pool.submit(write, os.path.join(sdir, "reads.all.pileup"), host))
            dirs.append(sdir)
            total += len(host)
            if len(futures) > 16:
                futures.pop(0).result()
        for fu in futures:
            fu.result()
    dirs_file = os.path.join(tmpdir, "sampleDirectories.txt")
    with open(dirs_file, "w") as f:
        f.write("\n".join(dirs) + "\n")
    return tmpdir, ref_path, dirs_file, dirs, total


_H2D_PROBE = {}


def pinned_h2d_gbps(torch, n=256 << 20, reps=8):
    """What the host link gives pinned memory on THIS box: the best of several shapes of the copy — one, two and four streams each
    with its own pinned buffer (the streamed ingestion keeps several copies in flight; a single stream was measured at HALF the link
    rate on one of the driver's boxes, so a single-stream probe is no ceiling), buffers of 16 MiB to 256 MiB, allocated by this
    thread as the library's staging buffers are.  Measured once per process; `pinned_h2d_probe()` says which shape won."""
    if "best" in _H2D_PROBE:
        return _H2D_PROBE["best"]
    best, shapes = 0.0, []
    for streams in (1, 2, 4):
        for size in (16 << 20, 64 << 20, n):
            try:
                srcs = [torch.empty(size, dtype=torch.uint8, pin_memory=True) for _ in range(streams)]
                dsts = [torch.empty(size, dtype=torch.uint8, device="cuda") for _ in range(streams)]
                qs = [torch.cuda.Stream() for _ in range(streams)]
            except RuntimeError:
                continue
            rounds = max(2, min(reps * n // size // streams, 64))
            for k in range(streams):                                           # warm-up: the mappings, the engines
                with torch.cuda.stream(qs[k]):
                    dsts[k].copy_(srcs[k], non_blocking=True)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(rounds):
                for k in range(streams):
                    with torch.cuda.stream(qs[k]):
                        dsts[k].copy_(srcs[k], non_blocking=True)
            torch.cuda.synchronize()
            rate = rounds * streams * size / (time.perf_counter() - t) / 1e9
            shapes.append({"streams": streams, "buffer_bytes": size, "gb_per_sec": rate})
            if rate > best:
                best = rate
                _H2D_PROBE["shape"] = shapes[-1]
            del srcs, dsts, qs
    _H2D_PROBE["best"] = best
    _H2D_PROBE["shapes"] = shapes
    return best


def pinned_h2d_probe():
    return {"best": _H2D_PROBE.get("shape"), "all": _H2D_PROBE.get("shapes")}


def over_link(rate_gbps, h2d):
    """A path's rate as a fraction of the probed link rate — or None (and a flag) when the path beat the probe by more than 5 %:
    then the probe was no ceiling on this box and the ratio would say nothing."""
    if not h2d or rate_gbps > 1.05 * h2d:
        return None
    return rate_gbps / h2d


def pipeline_from_files(pile, offs, sizes, refh, G, n_files, with_separate=True):
    """The rank's shard as one job from files to files.  The sample tree is written first (page cache, then os.sync()); a two-sample job
    warms the process up (code objects, Python imports); then ONE run of hot_path_batch over all samples is timed — its first
    and only one, as in a real job (a second run in the same process would start by waiting for the driver to take back the
    54 GB the first one freed).  Beside it: bytes / pinned copy rate, and the separate subcommands on the same tree."""
    import shutil
    import torch
    from snp_pipeline_amd import hot_path
    need = int(sum(sizes[:n_files])) + (1 << 30)
    base_dir = None
    for cand in (__import__("tempfile").gettempdir(), "/dev/shm"):
        try:
            if shutil.disk_usage(cand).free > need + (8 << 30):
                base_dir = cand
                break
        except OSError:
            pass
    if base_dir is None:
        return {"skipped": "no room for %d bytes of pileup files" % need}
    # Device memory in its steady state: on a box fresh from boot the FIRST allocation of a stretch of device memory costs ~30 ms
    # per GiB (64 GiB: 1.9 s) and the copies that run beside it drop to half their rate; memory that has been allocated and freed
    # once comes back in microseconds (tools/pipeline_time.py --recycle).  A node that has run one job before is in that state.
    t0 = time.perf_counter()
    recycled = torch.empty(need + (4 << 30), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t_recycle = time.perf_counter() - t0
    n_recycled = recycled.numel()
    del recycled
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    tmpdir, ref_path, dirs_file, dirs, total = write_sample_tree(
        base_dir, refh, G, lambda i: pile[int(offs[i]):int(offs[i]) + sizes[i]].cpu().numpy(), n_files)
    t_tree = time.perf_counter() - t0
    # the job reads files that sit in the page cache and have been written back, as pileups made some time before would be: without
    # this the kernel's write-back of the 54 GB this process has just written runs beside the timed job (+0.1-0.2 s of reading)
    t0 = time.perf_counter()
    os.sync()
    t_sync = time.perf_counter() - t0
    try:
        warm = os.path.join(tmpdir, "warmup")
        os.makedirs(warm)
        with open(os.path.join(warm, "sampleDirectories.txt"), "w") as f:
            f.write("\n".join(dirs[:2]) + "\n")
        run_cli(hot_path_line(os.path.join(warm, "sampleDirectories.txt"), ref_path, " --workDir %s" % warm))
        h2d = pinned_h2d_gbps(torch)
        wall = run_cli(hot_path_line(dirs_file, ref_path))
        st = dict(hot_path.hot_path_batch.last_stats)
        ideal = total / (h2d * 1e9)
        out = {
            "what": "%d samples (%s): reads.all.pileup files in the page cache -> var.flt.vcf, var.flt_preserved/_removed.vcf, both snplists, "
                    "consensus(.fasta|.vcf) x 2 flows, snpma x 2, referenceSNP x 2, distance TSVs x 4 — one hot_path_batch job" % (n_files, base_dir),
            "samples": n_files, "pileup_bytes": total, "seconds": st["seconds"], "cli_seconds": wall,
            "h2d_bytes": st["h2d_bytes"], "each_pileup_crossed_the_link_once": st["h2d_bytes"] == total,
            "pinned_h2d_gb_per_sec": h2d, "bytes_over_pinned_h2d_seconds": ideal,
            "wall_over_copy_time": (st["seconds"] / ideal) if st["seconds"] >= ideal / 1.05 else None,
            "pinned_h2d_probe_is_a_ceiling_here": st["seconds"] >= ideal / 1.05, "pinned_h2d_probe": pinned_h2d_probe(),
            "samples_per_sec": n_files / st["seconds"], "pileup_gb_per_sec": total / st["seconds"] / 1e9,
            "phases_seconds": st["phases"], "ingest": st["ingest"], "readers": st.get("readers"), "usable_cores": st.get("usable_cores"),
            "local_world": st.get("local_world"), "snp_sites": st["sites"], "snp_sites_preserved": st["sites_preserved"],
            "tree_written_in_seconds": t_tree, "tree_synced_in_seconds": t_sync,
            "device_memory_recycled_first": {"bytes": n_recycled, "first_allocation_seconds": t_recycle,
                                             "note": "allocated and freed once before the tree was written: the job's own allocations "
                                                     "then take microseconds, as on a node that has run a job before"},
        }
        # the same job with the var.flt.vcf files as INPUTS (--siteCalling existing: what a tree whose site calling was done by the
        # reference's VarScan looks like): nothing under samples/*/var.flt.vcf may change, every other output must come out the same
        mine = output_digests(tmpdir, dirs)
        stamps = [os.stat(os.path.join(sdir, "var.flt.vcf")).st_mtime_ns for sdir in dirs]
        torch.cuda.empty_cache()
        run_cli(hot_path_line(dirs_file, ref_path, " --siteCalling existing"))
        st2 = dict(hot_path.hot_path_batch.last_stats)
        again = output_digests(tmpdir, dirs)
        out["site_calling_existing"] = {
            "seconds": st2["seconds"], "samples_per_sec": n_files / st2["seconds"], "mode": st2["site_calling"],
            "var_flt_vcf_untouched": stamps == [os.stat(os.path.join(sdir, "var.flt.vcf")).st_mtime_ns for sdir in dirs],
            "outputs_identical_to_the_device_route": again == mine, "phases_seconds": st2["phases"],
            "note": "second job in this process: it starts on the device memory the first one has just freed (bench.py pipeline_from_files)"}
        out["site_calling_mode"] = st["site_calling"]
        if again != mine or not out["site_calling_existing"]["var_flt_vcf_untouched"]:
            raise SystemExit("hot_path_batch --siteCalling existing: outputs differ from the device route, or var.flt.vcf was touched")
        if with_separate:
            for sdir in dirs:                                   # nothing of the one-job run is left to be "fresh"
                for name in PER_SAMPLE_FILES:
                    os.remove(os.path.join(sdir, name))
            sep = separate_steps(tmpdir, ref_path, dirs_file)
            theirs = output_digests(tmpdir, dirs)
            out["separate_steps_seconds"] = sep
            out["separate_steps_total_seconds"] = sum(sep.values())
            out["speedup_over_separate_steps"] = sum(sep.values()) / st["seconds"]
            out["outputs_identical_to_separate_steps"] = mine == theirs
            if mine != theirs:
                raise SystemExit("hot_path_batch and the separate subcommands disagree on %r" % [k for k in mine if mine[k] != theirs[k]])
        return out
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)


def side_row(fn, *a):
    """A side row of the bench line.  What the box cannot give it (room for the files, host memory, a writable scratch directory)
    shows up as {"error": ...} in that row instead of costing the run its headline; a RESULT that disagrees with its check ends the
    run as before (those raise SystemExit)."""
    try:
        return fn(*a)
    except Exception as err:                                    # noqa: B902
        import traceback
        return {"error": "%s: %s" % (type(err).__name__, err), "traceback": traceback.format_exc().splitlines()[-6:]}


def scan_shapes(d, L, dev, ref, alt, G, pos, n_samples):
    """The pileup-scan kernel on the shapes where it is weakest, measured the same way as the headline
    (HIP events around the launches of one batched call, 3 launches after a warm-up, as many samples per launch as the headline's
    shard has — a launch of a quarter of the bytes pays the same ramp and tail and reads 2-3 points lower): shallow pileups (more, shorter lines per
    tile), a deep one, CR LF line ends, and samples of many short contigs with long names.  Each entry: bytes per launch,
    GB/s, fraction of the HBM peak."""
    import torch
    out = {}
    prm = dev.make_params(0, 0.6, 3, 0, 0.0)
    S = len(pos)

    def measure(ss, buf, offs, sizes, n_sites):
        B = len(sizes)
        bases = torch.empty((B, max(n_sites, 1)), dtype=torch.uint8, device="cuda")
        filt = torch.empty((B, max(n_sites, 1)), dtype=torch.uint8, device="cuda")
        status = torch.empty((B, 4), dtype=torch.int64, device="cuda")
        run = lambda: d.call_consensus_batch_dev(ss, buf.data_ptr(), np.asarray(offs, dtype=np.uint64), prm, bases.data_ptr(), filt.data_ptr(),   # noqa: E731
                                                 status.data_ptr(), sizes=np.asarray(sizes, dtype=np.uint64))
        run()
        torch.cuda.synchronize()
        d.kernel_timing(True)
        d.kernel_time_ms(0)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ms, n = d.kernel_time_ms(0)
        d.kernel_timing(False)
        st = status.cpu().numpy()
        if (st[:, 0] != -1).any():
            raise SystemExit("scan_shapes: the scan reported a malformed pileup")
        nbytes = int(sum(sizes))
        gbps = nbytes / (ms / n * 1e-3) / 1e9
        return {"samples": B, "bytes_per_launch": nbytes, "avg_launch_ms": ms / n, "gb_per_sec": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBS,
                "lines_per_sample": int(st[0, 1]), "bases": bases}

    ss1 = d.siteset([(b"synth_chr1", int(p)) for p in pos], [L.SITE_IN_SNPLIST] * S)
    for label, depth, B in (("depth_8x", 8.0, n_samples), ("depth_15x", 15.0, n_samples), ("depth_100x", 100.0, max(1, n_samples * 3 // 8))):   # (100x: the headline's bytes)
        sizes = [d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), 0, 0, mean_depth=depth) for i in range(B)]
        offs = np.concatenate(([0], np.cumsum([(n + 255) // 256 * 256 for n in sizes])))
        buf = torch.empty(int(offs[-1]) + 8192, dtype=torch.uint8, device="cuda")
        for i in range(B):
            d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), buf.data_ptr() + int(offs[i]), sizes[i], mean_depth=depth)
        torch.cuda.synchronize()
        r = measure(ss1, buf, offs[:-1], sizes, S)
        r.pop("bases")
        out[label] = r
        del buf
    # CR LF: one 30x sample with "\r" put in front of every "\n" (torch index arithmetic on the device), replicated
    n = d.synth_pileup_dev(3, 0, G, ref.data_ptr(), alt.data_ptr(), 0, 0, mean_depth=30.0)
    lf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    d.synth_pileup_dev(3, 0, G, ref.data_ptr(), alt.data_ptr(), lf.data_ptr(), n, mean_depth=30.0)
    torch.cuda.synchronize()
    lf = lf[:n]
    is_nl = lf == 10
    before = torch.cumsum(is_nl.to(torch.int32), 0, dtype=torch.int64)            # '\n' at or before i
    crlf = torch.full((n + int(before[-1]),), 13, dtype=torch.uint8, device="cuda")
    crlf[torch.arange(n, device="cuda") + before] = lf                           # byte i moves behind the '\r's of the '\n's up to and including it
    del before, is_nl
    B = n_samples
    res = {}
    for label, one in (("lf_same_sample", lf), ("cr_lf", crlf)):
        step = (one.numel() + 255) // 256 * 256
        buf = torch.empty(B * step + 8192, dtype=torch.uint8, device="cuda")
        for i in range(B):
            buf[i * step:i * step + one.numel()] = one
        r = measure(ss1, buf, np.arange(B) * step, [one.numel()] * B, S)
        res[label] = bytes(r.pop("bases")[0].cpu().numpy())
        if label == "cr_lf":
            out[label] = r
        del buf
    if res["cr_lf"] != res["lf_same_sample"]:
        raise SystemExit("scan_shapes: CR LF consensus differs from the LF one")
    out["cr_lf"]["same_consensus_as_lf"] = True
    del lf, crlf
    # many contigs: 40 x 125 kbp per sample, names of 6 to 20 bytes, every fifth contig without a site
    C, Gc = 40, 125_000
    refc = torch.empty(Gc + 1, dtype=torch.uint8, device="cuda")
    d.synth_reference_dev(1, Gc, refc.data_ptr())
    posc = np.sort(np.random.default_rng(2).choice(np.arange(51, Gc - 49), size=Gc // 100, replace=False))
    alt_h = np.zeros(Gc + 1, dtype=np.uint8)
    alt_h[posc] = ord("A")
    altc = torch.from_numpy(alt_h).cuda()
    names = [("NODE_%d_len_%d" % (c + 1, Gc)).encode() if c % 2 else ("ctg%03d" % c).encode() for c in range(C)]
    keys = [(names[c], int(p)) for c in range(C) if c % 5 != 3 for p in posc]
    B = n_samples
    piece = [[d.synth_pileup_dev(3, s * C + c, Gc, refc.data_ptr(), altc.data_ptr(), 0, 0, contig=names[c]) for c in range(C)] for s in range(B)]
    total = sum(sum(x) for x in piece)
    buf = torch.empty(total + 8192, dtype=torch.uint8, device="cuda")
    offs, lens, o = [], [], 0
    for s_ in range(B):
        offs.append(o)
        for c in range(C):
            o += d.synth_pileup_dev(3, s_ * C + c, Gc, refc.data_ptr(), altc.data_ptr(), buf.data_ptr() + o, piece[s_][c], contig=names[c])
        lens.append(o - offs[-1])
    ssc = d.siteset(keys, [L.SITE_IN_SNPLIST] * len(keys))
    r = measure(ssc, buf, offs, lens, len(keys))
    r.pop("bases")
    out["contigs_40_x_125kbp"] = r
    return out


def _event_ms(torch, fn, reps=3):
    """Device time of fn() (enqueued on torch's current stream) by events, best of reps."""
    best = None
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ms = a.elapsed_time(b)
        best = ms if best is None or ms < best else best
    return best


def aux_steps(d, pos, G):
    """K3 / K4 at configs[3] scale (1000 samples x 1500 SNP records): device time of the _dev forms (device pointers, no
    host synchronisation inside) and wall time of the host-pointer forms (staging + kernels + read-back)."""
    import torch
    arng = np.random.default_rng(5)
    n_s, per = 1000, min(1500, len(pos))
    samp_pos = [np.sort(arng.choice(pos, size=per, replace=False)) for _ in range(n_s)]
    keys = np.concatenate(samp_pos).astype(np.int64)                # contig 0
    who = np.repeat(np.arange(n_s, dtype=np.int32), per)
    m = len(keys)
    tk, tw = torch.from_numpy(keys).cuda(), torch.from_numpy(who).cuda()
    ou = torch.zeros(m, dtype=torch.int64, device="cuda")
    oo = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
    oc = torch.zeros(m, dtype=torch.int32, device="cuda")
    on = torch.zeros(4, dtype=torch.int32, device="cuda")
    k_merge = _event_ms(torch, lambda: d.merge_sites_dev(tk.data_ptr(), tw.data_ptr(), m, ou.data_ptr(), oo.data_ptr(), oc.data_ptr(), on.data_ptr()))
    n_unique = int(on[0])
    seg = torch.arange(0, m + 1, per, dtype=torch.int32, device="cuda")
    cap = 3 * m
    ws_, we_ = torch.zeros(cap, dtype=torch.int64, device="cuda"), torch.zeros(cap, dtype=torch.int64, device="cuda")
    wg_, wn = torch.zeros(cap, dtype=torch.int32, device="cuda"), torch.zeros(2, dtype=torch.int32, device="cuda")
    k_dense = _event_ms(torch, lambda: d.dense_windows_dev(tk.data_ptr(), seg.data_ptr(), n_s, m, [3, 2, 1], [1000, 125, 15],
                                                          ws_.data_ptr(), we_.data_ptr(), wg_.data_ptr(), wn.data_ptr()))
    n_win = int(wn[0])
    grp = torch.zeros(max(n_win, 1), dtype=torch.int32, device="cuda")
    og = torch.zeros(max(n_win, 1), dtype=torch.int32, device="cuda")
    os_, oe = torch.zeros(max(n_win, 1), dtype=torch.int64, device="cuda"), torch.zeros(max(n_win, 1), dtype=torch.int64, device="cuda")
    mn = torch.zeros(2, dtype=torch.int32, device="cuda")
    k_mreg = _event_ms(torch, lambda: d.merge_regions_dev(grp.data_ptr(), ws_.data_ptr(), we_.data_ptr(), n_win, og.data_ptr(), os_.data_ptr(),
                                                          oe.data_ptr(), mn.data_ptr()))
    n_reg = int(mn[0])
    roff = torch.tensor([0, n_reg], dtype=torch.int32, device="cuda")
    pg = torch.zeros(m, dtype=torch.int32, device="cuda")
    flag = torch.zeros(m, dtype=torch.uint8, device="cuda")
    k_inreg = _event_ms(torch, lambda: d.in_regions_dev(pg.data_ptr(), tk.data_ptr(), m, roff.data_ptr(), os_.data_ptr(), oe.data_ptr(), 1, flag.data_ptr()))
    inside = int(flag.sum().item())
    # host-pointer forms
    hk, hw = keys.astype(np.uint64), who.astype(np.uint32)
    d.merge_sites(hk[:1000], hw[:1000])
    t1 = time.perf_counter()
    uniq, _, _ = d.merge_sites(hk, hw)
    t_merge = time.perf_counter() - t1
    hseg = np.arange(0, m + 1, per, dtype=np.uint32)
    t1 = time.perf_counter()
    hs, he, _ = d.dense_windows(keys, hseg, [3, 2, 1], [1000, 125, 15])
    t_dense = time.perf_counter() - t1
    t1 = time.perf_counter()
    _, rs_, re_ = d.merge_regions(np.zeros(len(hs), np.uint32), hs, he)
    t_mreg = time.perf_counter() - t1
    t1 = time.perf_counter()
    hin = d.in_regions(np.zeros(m, np.uint32), keys, [0, len(rs_)], rs_, re_)
    t_inreg = time.perf_counter() - t1
    assert (len(uniq), len(hs), len(rs_), int(hin.sum())) == (n_unique, n_win, n_reg, inside)
    return {
        "workload": "%d samples x %d SNP records each, one contig of %d bp" % (n_s, per, G),
        "device_ms": {"merge_sites_union_and_carriers": k_merge, "dense_windows_3_rules": k_dense, "merge_regions": k_mreg,
                      "in_regions": k_inreg,
                      "note": "_dev entry points: device pointers in and out, HIP events around the whole step, no host synchronisation inside"},
        "host_form_wall_ms": {"merge_sites_union_and_carriers": t_merge * 1e3, "dense_windows_3_rules": t_dense * 1e3,
                              "merge_regions": t_mreg * 1e3, "in_regions": t_inreg * 1e3,
                              "note": "host-pointer entry points: pageable H2D + kernels + D2H, one synchronisation at the end"},
        "unique_sites": n_unique, "windows": n_win, "regions": n_reg, "records_in_a_region": inside,
    }


def cpu_baseline(args, d, pile, offs, sizes, bases, pos, G, S, gpu_value, secondary):
    """The CPU oracle — a statement-for-statement Python port of the reference's loops, pinned to the reference by the
    golden vectors — on samples of the same batch.  Three legs, as BASELINE.md 3 plans: one core; one process per sample
    on min(cores, 32) cores (what run.py:709-710 / xargs -P do); the distance loop (single process in the reference)."""
    from oracle import pileup_oracle as po
    from oracle import steps_oracle as so
    B = len(sizes)
    ncpu = min(args.cpu_samples, B)
    snps = [(b"synth_chr1", int(p)) for p in pos]
    p = po.CallerParams(0, 0.6, 3, 0, 0.0)
    gpu_rows = bases[:max(ncpu, 1)].cpu().numpy()
    t_cpu = 0.0
    ok = True
    for i in range(ncpu):
        data = bytes(pile[int(offs[i]):int(offs[i]) + sizes[i]].cpu().numpy())
        t1 = time.perf_counter()
        cons, _ = po.call_consensus_sites(data, snps, set(), p)
        t_cpu += time.perf_counter() - t1
        ok = ok and (cons == bytes(gpu_rows[i]))
    if not ok:
        raise SystemExit("GPU consensus differs from the CPU oracle")
    res = {
        "value": ncpu * S / t_cpu, "unit": "bases/s", "cores": 1, "kind": "port",
        "sample": "%d of the same synthetic samples (%d bp x %gx, %d sites each), call_consensus path only, pure-Python oracle"
                  % (ncpu, G, args.depth, S),
        "seconds": t_cpu, "genome_bp_per_sec": ncpu * G / t_cpu, "matches_gpu": True,
        "gpu_over_cpu_1core": gpu_value / (ncpu * S / t_cpu),
        "reference_probe": "BASELINE.md 2: the real reference measured 1.2e4 consensus bases/s and 1.19e6 genome-bp/s on 1 core "
                           "(survey container, 200 kbp synthetic pileup)",
    }
    # the reference runs one call_consensus process per sample (xargs -P / run.py:710): one oracle process per sample, files in
    # the page cache, at TWO process counts — half of and all of the CPUs this process may use (affinity mask and cgroup quota:
    # the bench box shows 256 CPUs and grants the time of 16), at most 128 — so that how the rate grows with the processes is
    # measured rather than asserted
    if not args.skip_cpu_parallel:
        import multiprocessing as mp
        import shutil
        import tempfile
        cores = os.cpu_count() or 1
        usable = effective_cores()
        n_files = max(1, min(32, B))
        base_dir = _scratch_dir(int(sum(sizes[:n_files])))
        if base_dir is not None:
            tmpdir = tempfile.mkdtemp(prefix="snpbench_cpu_", dir=base_dir)
            try:
                paths = []
                for i in range(n_files):
                    path = os.path.join(tmpdir, "s%d.pileup" % i)
                    with open(path, "wb") as f:
                        f.write(pile[int(offs[i]):int(offs[i]) + sizes[i]].cpu().numpy().tobytes())
                    paths.append(path)
                rows = bases[:n_files].cpu().numpy()
                ctx_mp = mp.get_context("spawn")                     # no fork of a process that holds a HIP context
                legs = []
                top_n = max(1, min(args.cpu_procs or usable, 128))
                counts = sorted({max(1, top_n // 2), top_n})
                for nproc in counts:
                    jobs = [(paths[k % n_files], [int(x) for x in pos]) for k in range(nproc)]      # one sample per process
                    t1 = time.perf_counter()
                    with ctx_mp.Pool(nproc) as pool:
                        got = pool.map(_oracle_worker, jobs, chunksize=1)
                    t_par = time.perf_counter() - t1
                    same = bool(all(r == bytes(rows[k % n_files]) for k, r in enumerate(got)))
                    legs.append({"processes": nproc, "seconds": t_par, "value": nproc * S / t_par, "matches_gpu": same})
            finally:
                shutil.rmtree(tmpdir, ignore_errors=True)
            top = legs[-1]
            res["parallel"] = {
                "value": top["value"], "unit": "bases/s", "processes": top["processes"], "host_cores": cores, "usable_cores": usable,
                "seconds": top["seconds"],
                "matches_gpu": bool(all(leg["matches_gpu"] for leg in legs)), "legs": legs,
                "gpu_over_cpu": gpu_value / top["value"],
                "note": "one oracle process per sample incl. process start and file read, as the reference's xargs -P does; "
                        "legs = the same at every process count tried (the rate per process is value / processes)",
            }
            if len(legs) > 1:
                res["parallel"]["measured_scaling"] = {"processes": [leg["processes"] for leg in legs],
                                                       "rate_ratio": legs[-1]["value"] / legs[0]["value"],
                                                       "ideal_ratio": legs[-1]["processes"] / legs[0]["processes"]}
    # distance: the reference's per-pair Python loop (utils.py:1135-1165, distance.py:93-98), single process
    if args.cpu_dist_samples > 1:
        rng = np.random.default_rng(3)
        nd, sd = args.cpu_dist_samples, 50_000
        sym = rng.choice(np.frombuffer(b"ACGT-", dtype=np.uint8), size=(nd, sd), p=[.24, .24, .24, .24, .04]).astype(np.uint8)
        seqs = [bytes(r).decode() for r in sym]
        t1 = time.perf_counter()
        want = [[so.sequence_distance(seqs[i], seqs[j]) for j in range(i + 1, nd)] for i in range(nd)]
        t_d = time.perf_counter() - t1
        got = d.distance(sym)
        same = all(got[i, j] == want[i][j - i - 1] for i in range(nd) for j in range(i + 1, nd))
        if not same:
            raise SystemExit("GPU distances differ from the CPU oracle")
        pairs = nd * (nd - 1) / 2
        res["distance"] = {
            "value": pairs / t_d, "unit": "pairs/s", "cores": 1, "site_compares_per_sec": pairs * sd / t_d, "seconds": t_d,
            "sample": "%d x %d random ACGT- matrix, all pairs, pure-Python oracle" % (nd, sd), "matches_gpu": True,
            "gpu_over_cpu_site_compares": (secondary["site_compares_per_sec"] / (pairs * sd / t_d)) if secondary else None,
            "reference_probe": "BASELINE.md 2: the real reference measured 1.1e7 site-compares/s on 1 core",
        }
    return res


def secondary_distance(d, sharding, args, rank, world, multi, one_gpu, barrier, watch):
    """The distance step alone at configs[4] shape (kernel + row-band exchange when N > 1): pairs/s, site-compares/s, the integer
    VALU fraction, and a checksum of the rows this rank owns summed over ranks (the same numbers whatever the world size)."""
    import torch
    import torch.distributed as dist
    n2, s2 = args.dist_samples, args.dist_sites
    b2 = sharding.RowBands(n2, world)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    lut = torch.tensor(list(b"ACGT-"), dtype=torch.uint8, device="cuda")
    probs = torch.tensor([.24, .24, .24, .24, .04], device="cuda")
    sym = torch.empty((n2, s2), dtype=torch.uint8, device="cuda")
    chunk = max(1, (1 << 28) // s2)
    for r0 in range(0, n2, chunk):
        r1 = min(n2, r0 + chunk)
        idx = torch.multinomial(probs, (r1 - r0) * s2, replacement=True, generator=g)
        sym[r0:r1] = lut[idx].view(r1 - r0, s2)
        del idx
    pk = torch.zeros((b2.n_padded, d.packed_row_bytes(s2)), dtype=torch.uint8, device="cuda")
    d.pack_matrix_dev(sym.data_ptr(), n2, s2, s2, pk.data_ptr())
    del sym
    dm = torch.zeros((b2.n_padded, b2.n_padded), dtype=torch.int32, device="cuda")

    def dstep():
        d.distance_packed_dev(pk.data_ptr(), b2.n_padded, s2, dm.data_ptr(), rank, world)
        return b2.exchange(dm, rank) if multi else dm

    watch.phase("secondary: distance tiles + row-band exchange (warm-up)")
    dstep()                                                  # warm-up
    barrier()
    d.kernel_timing(True)
    d.kernel_time_ms(2)
    t1 = time.perf_counter()
    watch.phase("secondary: distance tiles + row-band exchange")
    for _ in range(args.dist_reps):
        band2 = dstep()
    barrier()
    el2 = (time.perf_counter() - t1) / args.dist_reps
    k_ms, k_n = d.kernel_time_ms(2)
    d.kernel_timing(False)
    if multi:
        tt = torch.tensor([el2], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el2 = float(tt.item())
    # what the rows this rank owns add up to, by row and by column (summed over ranks: the same numbers whatever the world
    # size — the strong-scaling test compares them with the one-rank run)
    lo2, hi2 = b2.band_rows(rank)
    mine2 = band2[:hi2 - lo2, :n2] if multi else band2[lo2:hi2, :n2]
    w_row = torch.arange(lo2, hi2, dtype=torch.int64, device="cuda") * 1000003 + 17
    w_col = torch.arange(n2, dtype=torch.int64, device="cuda") * 10007 + 3
    m64 = mine2.to(torch.int64)
    chk = torch.stack([m64.sum(), (m64.sum(dim=1) * w_row).sum(), (m64.sum(dim=0) * w_col).sum()])
    del m64
    if multi:
        chk_t = chk.cpu() if one_gpu else chk
        dist.all_reduce(chk_t, op=dist.ReduceOp.SUM)
        chk = chk_t
    band_checksum = [int(x) for x in chk.cpu().tolist()]
    del band2
    pairs = n2 * (n2 - 1) / 2
    # 4 VALU lane-ops per 32 site-compares (v_xor, 2 x v_bitop3, v_bcnt); integer VALU peak = 256 CU x 4 SIMD x 16 lanes
    # x 2.4 GHz
    valu_peak = 256 * 64 * 2.4e9 * world
    return {
        "metric": "pairwise_snp_distances_per_sec", "value": pairs / el2, "unit": "pairs/s",
        "site_compares_per_sec": pairs * s2 / el2, "seconds": el2,
        "config": {"workload": "BASELINE configs[4] shape: %d samples x %d sites, random ACGT- matrix; tiles dealt to %d rank(s)%s"
                               % (n2, s2, world, ", row-band exchange included" if multi else "")},
        "kernel_ms": k_ms / max(k_n, 1), "band_checksum": band_checksum,
        "valu_frac_of_peak": (pairs * s2 / 32 * 4 / el2) / valu_peak,
    }


def device_copy_gbps(torch):
    """A measured device-copy ceiling beside the spec peak: 1 GiB copied device to device, read + write bytes per second."""
    src = torch.empty(1 << 28, dtype=torch.int32, device="cuda")
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        dst.copy_(src)
    torch.cuda.synchronize()
    copy_s = (time.perf_counter() - t1) / 5
    return 2 * (1 << 30) / copy_s / 1e9


def committed_traffic(name, match):
    """A committed PMC result of the same workload (rocprofv3 cannot run inside this process): the newest profiles/<round>/<name> for
    which match(json) holds, as (json, "profiles/<round>/<name>"), else (None, None)."""
    for rnd in PROFILE_ROUNDS:
        try:
            with open(os.path.join(ROOT, "profiles", rnd, name)) as f:
                pt = json.load(f)
            if match(pt):
                return pt, "profiles/%s/%s" % (rnd, name)
        except (OSError, KeyError, ValueError, TypeError):
            pass
    return None, None


def live_traffic(args, algorithmic_bytes, timeout_s=240):
    """HBM traffic of one scan launch of THIS workload on THIS box, measured now: the timed step alone (a warm-up and three launches, no
    side rows) in two child processes under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, as
    MI355X_MICROARCH.md prescribes; counter collection cannot be switched on inside a running process).  FETCH_SIZE x 1024 x 2 (a
    128-byte line is tallied at 64 bytes: profiles/r6/fetch_calibration.md), WRITE_SIZE x 1024.  Returns the roofline's traffic fields,
    or {"error": ...} — the committed result of the same workload then stands in, and says so."""
    import csv
    import glob
    import shutil
    import subprocess
    import sys
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--samples", str(args.samples), "--genome", str(args.genome),
             "--sites", str(args.sites), "--depth", repr(args.depth), "--vcf-records", str(args.vcf_records), "--cpu-samples", "0", "--skip-secondary", "--skip-aux",
             "--e2e-files", "0", "--site-files", "0", "--pipeline-files", "0", "--shape-samples", "0", "--skip-call-variants", "--detail", "", "--no-live-traffic"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env["TMPDIR"] = "/tmp"
    got = {}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="snpbench_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", tmp, "--"] + child, cwd="/tmp", env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=timeout_s)
            vals = []
            for path in glob.glob(tmp + "/**/*counter_collection.csv", recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if row.get("Counter_Name") == counter and row.get("Kernel_Name", "").replace("(anonymous namespace)::", "").startswith("void k_scan_wave<false, 0>"):
                            vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or not vals:
                return {"error": "%s pass: exit code %d, %d launches seen; %s" % (counter, r.returncode, len(vals), r.stderr.decode("utf-8", "replace")[-300:])}
            got[counter] = (sum(vals) / len(vals), len(vals))
        except (OSError, subprocess.TimeoutExpired, ValueError, KeyError) as err:
            return {"error": "%s pass: %s: %s" % (counter, type(err).__name__, err)}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    fetch_b, write_b = got["FETCH_SIZE"][0] * 1024 * 2, got["WRITE_SIZE"][0] * 1024
    return {"traffic": fetch_b + write_b, "traffic_over_algorithmic": (fetch_b + write_b) / algorithmic_bytes if algorithmic_bytes else None,
            "traffic_fetch_bytes": fetch_b, "traffic_write_bytes": write_b, "traffic_launches_measured": got["FETCH_SIZE"][1],
            "traffic_source": "measured in this run: two child processes of the timed step under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; "
                              "FETCH_SIZE x 2 as calibrated in profiles/r6/fetch_calibration.md), %.0f s" % (time.perf_counter() - t0)}


def north_star(out, args, world, S):
    """The target of BASELINE.json's north_star, written down as numbers: 10 000 samples x 5 Mbp, call_consensus -> snp_matrix ->
    distance, reference CPU seconds over GPU seconds; >= 50 % of HBM peak on the scan; >= 6x at 8 GPUs."""
    ns = {"target": ">= 100x the reference CPU call_consensus -> snp_matrix -> distance throughput on 10 000 synthetic samples x 5 Mbp; "
                    ">= 50 % of HBM peak on the pileup scan at 1 GPU; >= 6x at 8 GPUs",
          "hbm_frac_of_peak_on_the_scan": out["roofline"]["frac"], "hbm_target": 0.5, "hbm_target_met": out["roofline"]["frac"] >= 0.5,
          "scaling_at_8_gpus": "unmeasured: no 8-GPU node was available to this build; the driver's SCALE run is the measurement"
                               if world == 1 else "this line is the %d-GPU point; the driver computes the ratio from its N = 1 line" % world}
    cb, sec, pipe = out.get("cpu_baseline"), out.get("secondary"), out.get("pipeline_from_files", {})
    if cb and sec and "distance" in cb and pipe.get("samples_per_sec"):
        n_s, bp, n_sites = 10_000, 5_000_000, 200_000
        par = cb.get("parallel")
        cpu_samples_per_sec = (par["value"] if par else cb["value"]) / S          # consensus of one 5 Mbp sample is scan-bound: per sample, not per site
        cpu_consensus_s = n_s / cpu_samples_per_sec
        cpu_distance_s = (n_s * (n_s - 1) / 2) * n_sites / cb["distance"]["site_compares_per_sec"]   # single process in the reference (distance.py:93-98)
        gpu_consensus_s = n_s / pipe["samples_per_sec"]                           # from files, the whole one-job path (site calling and both flows included)
        same_shape = (args.dist_samples, args.dist_sites) == (n_s, n_sites)
        gpu_distance_s = sec["seconds"] if same_shape else (n_s * (n_s - 1) / 2) * n_sites / sec["site_compares_per_sec"]
        file_ends_s = 1.4                                                         # snpma.fasta read + both TSVs written at that size (tools/distance_cli_time.py: 0.5-1.4 s)
        ns.update({
            "workload": "%d samples x %d bp x %gx, %d SNP sites" % (n_s, bp, args.depth, n_sites),
            "reference_cpu_seconds": {"consensus": cpu_consensus_s, "distance": cpu_distance_s, "total": cpu_consensus_s + cpu_distance_s,
                                      "how": "consensus: the CPU port with one process per sample on %s cores (%.3g samples/s, measured here on %s samples), x 10 000; "
                                             "distance: the per-pair Python loop at its measured %.3g site-compares/s, single process as in the reference"
                                             % ((par or {}).get("processes", 1), cpu_samples_per_sec, cb["sample"].split(" ")[0], cb["distance"]["site_compares_per_sec"])},
            "gpu_seconds_one_mi355x": {"consensus_from_files": gpu_consensus_s, "distance": gpu_distance_s, "distance_file_ends": file_ends_s,
                                       "total": gpu_consensus_s + gpu_distance_s + file_ends_s,
                                       "how": "consensus: hot_path_batch from pileup files at its measured %.1f samples/s (every pileup over the host link once; also "
                                              "site calling, region filter, both flows, VCFs), x 10 000; distance: %s"
                                              % (pipe["samples_per_sec"], "measured at this very shape" if same_shape else "scaled from the measured site-compare rate")},
            "ratio": (cpu_consensus_s + cpu_distance_s) / (gpu_consensus_s + gpu_distance_s + file_ends_s),
            "ratio_consensus_only": cpu_consensus_s / gpu_consensus_s, "ratio_distance_only": cpu_distance_s / (gpu_distance_s + file_ends_s),
            "ratio_target": 100.0,
            "note": "the reference's CPU seconds are extrapolated linearly from bounded samples (both steps are linear in their work); "
                    "the GPU consensus seconds start from files and are bound by the host link, not by the kernels"})
        ns["ratio_target_met"] = ns["ratio"] >= 100.0
    return ns



def from_files_ratios(out, S):
    """The CPU ratios for the rates that include the host link (files -> results), next to the HBM-resident ones."""
    cb = out["cpu_baseline"]
    e2e = out.get("end_to_end", {}).get("consensus_bases_per_sec")
    pipe = out.get("pipeline_from_files", {}).get("samples_per_sec")
    ratios = {"note": "GPU rates that start from FILES over the CPU oracle's call_consensus rate (the pipeline row also does site calling, "
                      "the region filter, both flows, matrices and distances in that time)"}
    if e2e:
        ratios["end_to_end_over_cpu_1core"] = e2e / cb["value"]
        if "parallel" in cb:
            ratios["end_to_end_over_cpu_parallel"] = e2e / cb["parallel"]["value"]
    if pipe:
        ratios["pipeline_from_files_over_cpu_1core"] = pipe * S / cb["value"]
        if "parallel" in cb:
            ratios["pipeline_from_files_over_cpu_parallel"] = pipe * S / cb["parallel"]["value"]
    cb["from_files"] = ratios

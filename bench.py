#!/usr/bin/env python3
"""bench.py — throughput of the post-alignment hot path on MI355X, one process per GPU.

A *step* is one pass of the hot path over one batch of synthetic samples already resident in HBM:
    call_consensus (pileup scan + per-site caller) for this rank's B samples
    -> pack the B x S consensus matrix 4 bits/site -> all-gather of packed rows over RCCL (N > 1)
    -> all-pairs SNP distance over the (N*B) x S matrix, 128x128 tiles dealt cyclically to ranks.
Workload at N = 1: BASELINE.json configs[1] size (the Agona set: ~25 samples, ~5 Mbp reference; B = 24 samples per
rank) with the synthetic pileups configs[3] specifies (30x depth, 50 k SNP sites) because the reference bundles no
pileup files.  N > 1 is weak scaling: B samples per rank (the node-level run of configs[3] would be 8 ranks x 125
samples, same per-sample work).  value = consensus bases called per second, whole job.

The same JSON line carries
  roofline      the pileup-scan kernel: algorithmic bytes (= pileup text bytes, each read once) / its average launch
                duration measured with HIP events recorded on the launch stream inside the timed region;
  cpu_baseline  the CPU oracle (a statement-for-statement Python port of the reference's loop structure) on the
                first samples of the same batch, 1 core, rank 0, N=1 only;
  secondary     pairwise SNP distances/s of the distance kernel alone at BASELINE configs[4] shape
                (10 000 x 200 000), timed separately after the K steps.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=24, help="samples per rank resident in HBM")
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--sites", type=int, default=50_000)
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--dist-samples", type=int, default=10_000)
    ap.add_argument("--dist-sites", type=int, default=200_000)
    ap.add_argument("--dist-reps", type=int, default=2)
    ap.add_argument("--cpu-samples", type=int, default=3, help="samples timed on the CPU oracle (0 = skip)")
    ap.add_argument("--skip-secondary", action="store_true")
    ap.add_argument("--skip-aux", action="store_true", help="skip the device-copy ceiling and the K3/K4 timings")
    ap.add_argument("--skip-cpu-parallel", action="store_true", help="skip the one-process-per-sample CPU baseline")
    ap.add_argument("--e2e-files", type=int, default=16, help="pileup files streamed from the page cache for the end_to_end row (0 = skip)")
    return ap.parse_args()


def _oracle_worker(job):
    """One call_consensus of the CPU oracle in its own process (bench cpu_baseline.parallel)."""
    path, positions = job
    from oracle import pileup_oracle as po
    with open(path, "rb") as f:
        data = f.read()
    cons, _ = po.call_consensus_sites(data, [(b"synth_chr1", p) for p in positions], set(), po.CallerParams(0, 0.6, 3, 0, 0.0))
    return cons


def end_to_end(d, ss, prm, pile, offs, sizes, bases, n_files, S):
    """Page-cache files -> FASTA bytes: the rate a run over more samples than fit in HBM proceeds at.  The files are
    written first (page cache / tmpfs), one warm-up file goes through (pinned staging allocation), then all of them are
    timed in one snpgpu_call_consensus_files call and compared with the resident results; the yardstick is a pinned
    host-to-device copy measured in the same process."""
    import shutil
    import tempfile
    import torch
    need = int(sum(sizes[:n_files])) + (64 << 20)
    base_dir = None
    # a regular file system first: the FIRST read of freshly written tmpfs pages is serialised in the kernel (~15 GB/s on
    # the bench box whatever the thread count; every later pass, and every pass over ordinary page-cache files, is not)
    for cand in (tempfile.gettempdir(), "/dev/shm"):
        try:
            if shutil.disk_usage(cand).free > 2 * need:
                base_dir = cand
                break
        except OSError:
            pass
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
        # pinned host -> device copy rate (the ceiling of this path)
        n = 256 << 20
        src = torch.empty(n, dtype=torch.uint8).pin_memory()
        dst = torch.empty(n, dtype=torch.uint8, device="cuda")
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(8):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        h2d = 8 * n / (time.perf_counter() - t1) / 1e9
        del src, dst
        d.call_consensus_files(ss, paths[:1], prm)                                  # warm-up: pinned staging, device slots
        res, rcs, st = d.call_consensus_files(ss, paths, prm)
        ok = all(int(rc) == 0 for rc in rcs) and all(bytes(res[i].bases) == bytes(bases[i].cpu().numpy()) for i in range(n_files))
        if not ok:
            raise SystemExit("streamed consensus differs from the resident one")
        gbps = st.bytes / st.seconds / 1e9
        return {
            "what": "%d pileup files in the page cache (%s) -> consensus bytes on the host, one snpgpu_call_consensus_files call"
                    % (n_files, base_dir),
            "files": n_files, "bytes": int(st.bytes), "seconds": st.seconds, "pileup_gb_per_sec": gbps,
            "consensus_bases_per_sec": n_files * S / st.seconds, "samples_per_sec": n_files / st.seconds,
            "pinned_h2d_gb_per_sec": h2d, "frac_of_pinned_h2d": gbps / h2d,
            "chunk_bytes": int(st.chunk_bytes), "reader_threads": int(st.n_readers), "staging_buffers": int(st.n_staging),
            "seconds_waiting_for_readers": st.seconds_waiting_for_readers,
            "seconds_waiting_for_device": st.seconds_waiting_for_device, "matches_resident": True,
        }
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from snp_pipeline_amd import _lib as L
    from snp_pipeline_amd import device as dev
    from snp_pipeline_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): launch with torch.distributed.run" % (world, args.gpus))
    # functional test hook (not a measurement mode): all ranks on one GPU over gloo, to exercise the N > 1 code path on
    # a single-GPU box
    if os.environ.get("SNPGPU_BENCH_TEST_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if os.environ.get("SNPGPU_BENCH_TEST_ONE_GPU") == "1":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    d = dev.Device(local_rank)
    d.use_torch_stream()

    G, S, B = args.genome, args.sites, args.samples
    # ---- synthetic inputs: reference seed 1, sites seed 2, pileups seed 3 -----------------------
    ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda")
    d.synth_reference_dev(1, G, ref.data_ptr())
    torch.cuda.synchronize()
    refh = ref.cpu().numpy()
    rng = np.random.default_rng(2)
    pos = np.sort(rng.choice(np.arange(501, G - 499, dtype=np.int64), size=S, replace=False))
    n_clustered = S // 100                                   # 1 % of the sites in clusters of 4 within 100 bp
    for c in range(0, n_clustered - 3, 4):
        base = pos[c * 25 % (S - 4)]
        pos[c:c + 4] = base + np.array([0, 17, 41, 83])
    pos = np.unique(np.clip(pos, 501, G - 500))
    S = len(pos)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    code = np.searchsorted(acgt, refh[pos])
    alt_host = np.zeros(G + 1, dtype=np.uint8)
    alt_host[pos] = acgt[(code + 1 + rng.integers(0, 3, size=S)) % 4]
    alt = torch.from_numpy(alt_host).cuda()

    sizes = []
    for i in range(B):
        sizes.append(d.synth_pileup_dev(3, rank * B + i, G, ref.data_ptr(), alt.data_ptr(), 0, 0, mean_depth=args.depth))
    offs = np.zeros(B + 1, dtype=np.uint64)
    for i, n in enumerate(sizes):
        offs[i + 1] = offs[i] + ((n + 255) // 256) * 256
    pile = torch.empty(int(offs[-1]) + 256, dtype=torch.uint8, device="cuda")
    for i in range(B):
        n = d.synth_pileup_dev(3, rank * B + i, G, ref.data_ptr(), alt.data_ptr(), pile.data_ptr() + int(offs[i]),
                               sizes[i], mean_depth=args.depth)
        assert n == sizes[i]
    torch.cuda.synchronize()
    pile_bytes = int(sum(sizes))

    keys = [(b"synth_chr1", int(p)) for p in pos]
    ss = d.siteset(keys, [L.SITE_IN_SNPLIST] * S)
    prm = dev.make_params(0, 0.6, 3, 0, 0.0)                  # pipeline defaults (snppipeline.conf:249)

    bases = torch.empty((B, S), dtype=torch.uint8, device="cuda")
    filt = torch.empty((B, S), dtype=torch.uint8, device="cuda")
    status = torch.empty((B, 4), dtype=torch.int64, device="cuda")
    row_bytes = d.packed_row_bytes(S)
    packed = torch.empty((B, row_bytes), dtype=torch.uint8, device="cuda")
    dmat = torch.zeros((world * B, world * B), dtype=torch.int32, device="cuda")

    sizes_np = np.asarray(sizes, dtype=np.uint64)

    def step():
        # one scan launch and one call launch for the rank's whole batch; sample i is bytes [offs[i], offs[i] + sizes[i])
        d.call_consensus_batch_dev(ss, pile.data_ptr(), offs[:B], prm, bases.data_ptr(), filt.data_ptr(), status.data_ptr(),
                                   sizes=sizes_np)
        d.pack_matrix_dev(bases.data_ptr(), B, S, S, packed.data_ptr())
        packed_all = sharding.all_gather_rows(packed, world * B)         # C2: RCCL all-gather over xGMI when world > 1
        d.distance_packed_dev(packed_all.data_ptr(), world * B, S, dmat.data_ptr(), rank, world)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    d.kernel_timing(True)
    d.kernel_time_ms(0), d.kernel_time_ms(1), d.kernel_time_ms(2)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    scan_ms, scan_n = d.kernel_time_ms(0)
    call_ms, call_n = d.kernel_time_ms(1)
    dist_ms, dist_n = d.kernel_time_ms(2)
    d.kernel_timing(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    st = status.cpu().numpy()
    if (st[:, 0] != -1).any():
        raise SystemExit("scan reported a malformed pileup: %r" % st[:, 0])
    if (filt.cpu().numpy() & 0x80).any():
        raise SystemExit("caller reported a malformed line")

    ms_per_step = elapsed * 1e3 / args.steps
    value = world * B * S / (elapsed / args.steps)
    scan_avg_ms = scan_ms / max(scan_n, 1)
    algo_bytes = pile_bytes * args.steps / max(scan_n, 1)       # per launch: the rank's whole batch of pileup text
    achieved = algo_bytes / (scan_avg_ms * 1e-3) / 1e9 if scan_avg_ms > 0 else 0.0

    # HBM traffic of one scan launch from the committed PMC passes (rocprofv3 cannot run inside this process); only
    # quoted when it was measured on this very workload
    traffic = None
    traffic_note = None
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1", "pmc_traffic.json")) as f:
            pt = json.load(f)
        w = pt["workload"]
        if (w["samples_per_gpu"], w["genome_bp"], w["mean_depth"], w["snp_sites"]) == (B, G, args.depth, S):
            traffic = pt["traffic_bytes_per_launch"]
            traffic_note = "profiles/r1/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
    except (OSError, KeyError, ValueError):
        pass

    out = {
        "metric": "consensus_bases_called_per_sec", "value": value, "unit": "bases/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1] size (Agona-like: %d samples/GPU x %d bp reference) with the synthetic "
                               "pileups configs[3] specifies (%gx depth, %d SNP sites; the reference bundles no pileups); "
                               "step = one batched scan launch + one call launch, 4-bit pack, all-gather, all-pairs distance"
                               % (B, G, args.depth, S),
                   "samples_per_gpu": B, "genome_bp": G, "mean_depth": args.depth, "snp_sites": S,
                   "pileup_bytes_per_gpu": pile_bytes, "caller": "q0 c0.6 D3 d0 b0",
                   "parallelism": "samples sharded, %d rank(s)" % world},
        "genome_bp_per_sec": world * B * G / (elapsed / args.steps),
        "pileup_gb_per_sec": world * pile_bytes / (elapsed / args.steps) / 1e9,
        "roofline": {"kernel": "k_scan_wave", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                     "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": scan_avg_ms, "launches": scan_n},
        "kernels_ms_per_step": {"k_scan_wave": scan_ms / args.steps, "k_call_sites": call_ms / args.steps,
                                "k_distance": dist_ms / args.steps},
    }

    # ---- secondary metric: the distance kernel alone at configs[4] shape ------------------------------------------
    if not args.skip_secondary:
        n2, s2 = args.dist_samples, args.dist_sites
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
        pk = torch.empty((n2, d.packed_row_bytes(s2)), dtype=torch.uint8, device="cuda")
        d.pack_matrix_dev(sym.data_ptr(), n2, s2, s2, pk.data_ptr())
        del sym
        dm = torch.zeros((n2, n2), dtype=torch.int32, device="cuda")
        d.distance_packed_dev(pk.data_ptr(), n2, s2, dm.data_ptr(), rank, world)      # warm-up
        barrier()
        d.kernel_timing(True)
        d.kernel_time_ms(2)
        t1 = time.perf_counter()
        for _ in range(args.dist_reps):
            d.distance_packed_dev(pk.data_ptr(), n2, s2, dm.data_ptr(), rank, world)
        barrier()
        el2 = (time.perf_counter() - t1) / args.dist_reps
        k_ms, k_n = d.kernel_time_ms(2)
        d.kernel_timing(False)
        if world > 1:
            tt = torch.tensor([el2], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el2 = float(tt.item())
        pairs = n2 * (n2 - 1) / 2
        # 4 VALU lane-ops per 32 site-compares (v_xor, 2 x v_bitop3, v_bcnt); integer VALU peak = 256 CU x 4 SIMD x 16 lanes
        # x 2.4 GHz
        valu_peak = 256 * 64 * 2.4e9
        out["secondary"] = {
            "metric": "pairwise_snp_distances_per_sec", "value": pairs / el2, "unit": "pairs/s",
            "site_compares_per_sec": pairs * s2 / el2, "seconds": el2,
            "config": {"workload": "BASELINE configs[4] shape: %d samples x %d sites, random ACGT- matrix" % (n2, s2)},
            "kernel_ms": k_ms / max(k_n, 1),
            "valu_frac_of_peak": (pairs * s2 / 32 * 4 / el2) / valu_peak,
        }
        del pk, dm

    # ---- context figures (rank 0, N = 1): a measured device-copy ceiling and the small latency-bound steps ------------
    if rank == 0 and world == 1 and not args.skip_aux:
        src = torch.empty(1 << 28, dtype=torch.int32, device="cuda")
        dst = torch.empty_like(src)
        dst.copy_(src)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            dst.copy_(src)
        torch.cuda.synchronize()
        copy_s = (time.perf_counter() - t1) / 5
        del src, dst
        out["roofline"]["measured_copy_gbps_read_plus_write"] = 2 * (1 << 30) / copy_s / 1e9
        # K3 / K4 at configs[3] scale: 1000 samples x ~1500 phase-1 SNP records each (host buffers in, host buffers out)
        arng = np.random.default_rng(5)
        n_s, per = 1000, 1500
        samp_pos = [np.sort(arng.choice(pos, size=per, replace=False)) for _ in range(n_s)]
        keys = np.concatenate(samp_pos).astype(np.uint64)              # contig 0
        who = np.repeat(np.arange(n_s, dtype=np.uint32), per)
        d.merge_sites(keys[:1000], who[:1000])                        # warm-up
        t1 = time.perf_counter()
        uniq, off_, car = d.merge_sites(keys, who)
        t_merge = time.perf_counter() - t1
        seg = np.arange(0, n_s * per + 1, per, dtype=np.uint32)
        t1 = time.perf_counter()
        ws_, we_, wg_ = d.dense_windows(keys.astype(np.int64), seg, [3, 2, 1], [1000, 125, 15])
        t_dense = time.perf_counter() - t1
        t1 = time.perf_counter()
        rg, rs_, re_ = d.merge_regions(np.zeros(len(ws_), np.uint32), ws_, we_)
        t_mreg = time.perf_counter() - t1
        t1 = time.perf_counter()
        inside = d.in_regions(np.zeros(len(keys), np.uint32), keys.astype(np.int64), [0, len(rs_)], rs_, re_)
        t_inreg = time.perf_counter() - t1
        out["aux_steps_ms"] = {
            "workload": "%d samples x %d SNP records each, one contig of %d bp" % (n_s, per, G),
            "merge_sites_union_and_carriers": t_merge * 1e3, "unique_sites": int(len(uniq)),
            "dense_windows_3_rules": t_dense * 1e3, "windows": int(len(ws_)),
            "merge_regions": t_mreg * 1e3, "regions": int(len(rs_)),
            "in_regions": t_inreg * 1e3, "records_in_a_region": int(inside.sum()),
            "note": "wall time of the host-buffer entry points (H2D + kernels + D2H); latency-bound, reported for completeness",
        }

    # ---- end to end: pileup FILES in the page cache -> consensus bytes on the host, through the streamed ingestion ----
    if rank == 0 and world == 1 and args.e2e_files > 0:
        out["end_to_end"] = end_to_end(d, ss, prm, pile, offs, sizes, bases, min(args.e2e_files, B), S)

    # ---- CPU baseline: the oracle on the first samples of the batch, one core, rank 0, N = 1 ---------------------
    if rank == 0 and world == 1 and args.cpu_samples > 0:
        from oracle import pileup_oracle as po
        ncpu = min(args.cpu_samples, B)
        snps = [(b"synth_chr1", int(p)) for p in pos]
        p = po.CallerParams(0, 0.6, 3, 0, 0.0)
        gpu_rows = bases[:ncpu].cpu().numpy()
        t_cpu = 0.0
        ok = True
        for i in range(ncpu):
            data = bytes(pile[int(offs[i]):int(offs[i]) + sizes[i]].cpu().numpy())
            t1 = time.perf_counter()
            cons, _ = po.call_consensus_sites(data, snps, set(), p)
            t_cpu += time.perf_counter() - t1
            ok = ok and (cons == bytes(gpu_rows[i]))
        out["cpu_baseline"] = {
            "value": ncpu * S / t_cpu, "unit": "bases/s", "cores": 1, "kind": "port",
            "sample": "%d of the same synthetic samples (%d bp x %gx, %d sites each), call_consensus path only, "
                      "pure-Python oracle" % (ncpu, G, args.depth, S),
            "seconds": t_cpu, "genome_bp_per_sec": ncpu * G / t_cpu, "matches_gpu": bool(ok),
        }
        if not ok:
            raise SystemExit("GPU consensus differs from the CPU oracle")
        # the reference runs one call_consensus process per sample (xargs -P / run.py:710): the same samples again, one
        # oracle process each, for the host's parallel rate
        if ncpu > 1 and not args.skip_cpu_parallel:
            import multiprocessing as mp
            import tempfile
            tmpdir = tempfile.mkdtemp(prefix="snpbench_")
            paths = []
            for i in range(ncpu):
                path = os.path.join(tmpdir, "s%d.pileup" % i)
                with open(path, "wb") as f:
                    f.write(bytes(pile[int(offs[i]):int(offs[i]) + sizes[i]].cpu().numpy()))
                paths.append(path)
            ctx_mp = mp.get_context("spawn")                     # no fork of a process that holds a HIP context
            t1 = time.perf_counter()
            with ctx_mp.Pool(ncpu) as pool:
                res = pool.map(_oracle_worker, [(pth, [int(x) for x in pos]) for pth in paths])
            t_par = time.perf_counter() - t1
            for pth in paths:
                os.remove(pth)
            os.rmdir(tmpdir)
            out["cpu_baseline"]["parallel"] = {
                "value": ncpu * S / t_par, "unit": "bases/s", "processes": ncpu, "host_cores": os.cpu_count(),
                "seconds": t_par, "matches_gpu": bool(all(r == bytes(gpu_rows[i]) for i, r in enumerate(res))),
                "note": "one oracle process per sample incl. process start and file read, as the reference's xargs -P does",
            }

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

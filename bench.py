#!/usr/bin/env python3
"""bench.py — throughput of the post-alignment hot path on MI355X, one process per GPU.

A *step* is one pass of the hot path over one batch of synthetic samples already resident in HBM:
    C1  all-gather of every rank's per-sample SNP records (site keys) + the site union on the device (merge_sites)
    call_consensus (pileup scan + per-site caller) for this rank's samples: one scan launch + one call launch
    -> pack the consensus matrix 4 bits/site -> C2 all-gather of packed rows over RCCL (N > 1)
    -> all-pairs SNP distance over the (N*B) x S matrix, 128x128 tiles dealt cyclically to ranks
    -> row-band exchange (all-to-all): every rank ends with the complete distance rows of its band.
Workload at N = 1: the per-GPU shard of BASELINE.json configs[3] — 125 samples (1000 / 8) x 5 Mbp x 30x synthetic
pileups (54 GB of text resident in HBM), 50 k SNP sites, 1 500 SNP records per sample.  N > 1: weak scaling by default
(125 samples per rank: N = 8 is configs[3] itself); `--scaling strong` keeps the total at --samples.  value = consensus
bases called per second, whole job.

Output: the LAST line of stdout is ONE compact JSON object (under 4 KB: the driver reads a bounded tail) with the contract's
keys, `roofline` (the pileup-scan kernel: algorithmic bytes = pileup text bytes, each read once, over its average launch
duration measured with HIP events on the launch stream inside the timed region) and `cpu_baseline` (the CPU oracle — a
statement-for-statement Python port of the reference's loops — on samples of the same batch; rank 0, N = 1).  Everything else
(the side rows of bench_rows.py, per-rank phases, probe tables, notes) goes to the detail file, `--detail`
(default gpurun_out/bench_detail.json), of which the compact line is an extract.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import bench_rows as rows                                   # noqa: E402
from bench_rows import HBM_PEAK_GBS                         # noqa: E402

COMPACT_LIMIT = 4096           # bytes: the last stdout line must stay under this (tests/test_gpu_tools.py)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=125, help="samples per rank (weak scaling) or in total (strong scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--sites", type=int, default=50_000)
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--vcf-records", type=int, default=1500, help="phase-1 SNP records per sample (input of the site merge, C1)")
    ap.add_argument("--dist-samples", type=int, default=10_000)
    ap.add_argument("--dist-sites", type=int, default=200_000)
    ap.add_argument("--dist-reps", type=int, default=2)
    ap.add_argument("--cpu-samples", type=int, default=8, help="samples timed on the CPU oracle, one core (0 = skip the CPU baseline)")
    ap.add_argument("--cpu-procs", type=int, default=0, help="one-process-per-sample CPU leg: processes (0 = min(cores, 32))")
    ap.add_argument("--cpu-dist-samples", type=int, default=200, help="rows of the CPU distance leg (x 50 000 sites; BASELINE.md 3: >= 200)")
    ap.add_argument("--skip-secondary", action="store_true")
    ap.add_argument("--skip-call-variants", action="store_true", help="leave out the with-counts / strict / --vcfAllPos rows")
    ap.add_argument("--site-files", type=int, default=16, help="pileup files for the site_calling row (0 = skip)")
    ap.add_argument("--skip-aux", action="store_true", help="skip the device-copy ceiling and the K3/K4 timings")
    ap.add_argument("--shape-samples", type=int, default=125, help="samples per launch of the scan_shapes rows, as in the headline's shard (0 = skip)")
    ap.add_argument("--skip-cpu-parallel", action="store_true", help="skip the one-process-per-sample CPU baseline")
    ap.add_argument("--e2e-files", type=int, default=16, help="pileup files streamed from the page cache for the end_to_end row (0 = skip)")
    ap.add_argument("--pipeline-files", type=int, default=125, help="samples of the pipeline_from_files row (0 = skip)")
    ap.add_argument("--skip-separate-steps", action="store_true", help="pipeline_from_files: do not time the separate subcommands beside it")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not measure the scan's HBM traffic with rocprofv3 child runs (the committed result of the same workload stands in)")
    ap.add_argument("--detail", type=str, default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="where rank 0 writes the full result (side rows, per-rank phases, notes); '' = nowhere")
    ap.add_argument("--dump", type=str, default=None, help="write this rank's results (site union, packed matrix, distance band) to DUMP.rankN.npz")
    return ap.parse_args()



class StepWatch(object):
    """A daemon thread that ends the process when one phase of a step takes longer than `limit` seconds: with a rank missing a
    collective never returns, and a run that hangs tells nobody where.  phase(None) disarms it."""

    def __init__(self, rank, limit):
        import threading
        self.rank, self.limit = rank, limit
        self.name, self.since = None, 0.0
        self.lock = threading.Lock()
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def phase(self, name):
        with self.lock:
            self.name, self.since = name, time.monotonic()

    def _run(self):
        while True:
            time.sleep(1.0)
            with self.lock:
                name, since = self.name, self.since
            if name is not None and time.monotonic() - since > self.limit:
                sys.stderr.write("bench.py rank %d: phase '%s' did not finish within %.0f s (a rank that never arrived?); giving up\n"
                                 % (self.rank, name, self.limit))
                sys.stderr.flush()
                os._exit(3)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start N ranks of this very command line on this node (one per GPU,
    rendezvous on 127.0.0.1 — the fan-out the reference does with its per-sample job arrays, run.py:613-627) and pass on what
    rank 0 prints.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:                                  # a free port: two benches on one node do not collide
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: what RCCL needs on this platform
    env.setdefault("OMP_NUM_THREADS", "8")
    env["SNPGPU_BENCH_LAUNCHER"] = "self"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def claim_stdout():
    """stdout carries the result line and nothing else: file descriptor 1 is pointed at stderr for the life of the process — RCCL
    prints "Librccl path : ..." on C stdio's stdout, which a pipe holds back until the process EXITS, i.e. after the JSON line, and
    every rank of a launch shares the pipe — and the returned descriptor is the real stdout, written once by rank 0 at the end."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return real


def open_rank(args):
    """This process as a rank of the job: device, process group (RCCL, or gloo in the one-GPU test hook), the library's context and —
    over RCCL — its communicator, the step watchdog.  Returns a namespace."""
    import types
    import torch
    import torch.distributed as dist
    from snp_pipeline_amd import device as dev
    from snp_pipeline_amd import sharding
    r = types.SimpleNamespace(torch=torch, dist=dist, dev=dev, sharding=sharding)
    r.rank = int(os.environ.get("RANK", "0"))
    r.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    r.world = int(os.environ.get("WORLD_SIZE", "1"))
    if r.world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): launch %d ranks with torch.distributed.run, or run plain "
                         "`python bench.py --gpus %d` and let it start them" % (r.world, args.gpus, args.gpus, args.gpus))
    # functional test hook (not a measurement mode): all ranks on one GPU over gloo, to exercise the N > 1 code path on
    # a single-GPU box
    r.one_gpu = os.environ.get("SNPGPU_BENCH_TEST_ONE_GPU") == "1"
    if r.one_gpu:
        r.local_rank = 0
    torch.cuda.set_device(r.local_rank)
    r.backend = None
    # (a second functional hook: SNPGPU_DIST_AT_WORLD_1=1 makes a group of one rank and still runs every collective of the
    # N > 1 step — the RCCL calls on device tensors — on a box with one GPU)
    r.multi = r.world > 1 or sharding.group_of_one_exchanges()
    if r.multi:
        r.backend = "gloo" if r.one_gpu else "nccl"
        if r.one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", r.local_rank))
    r.d = dev.Device(r.local_rank)
    r.d.use_torch_stream()
    # over RCCL the exchanges of the step are calls into libsnpgpu.so on the context's stream (csrc/comm.hip: ncclAllGather, grouped
    # ncclSend / ncclRecv, hand-written tile kernels); torch.distributed stays for the barrier and for adding up the timings.  The
    # gloo hook (all ranks on one GPU) and SNPGPU_COMM=torch keep the torch.distributed route for the exchanges too.
    r.abi_route = bool(r.multi and not r.one_gpu and os.environ.get("SNPGPU_COMM") != "torch" and sharding.use_abi_comm(r.d))
    r.comm_info = r.d.comm_info() if r.abi_route else None
    # a rank that never arrives must end the run, not hang it: every step is watched (the phase it was in goes to stderr, exit code 3)
    r.watch = StepWatch(r.rank, float(os.environ.get("SNPGPU_BENCH_STEP_TIMEOUT", "600")))

    def barrier():
        r.watch.phase("barrier")
        torch.cuda.synchronize()
        if r.multi:
            dist.barrier()
        torch.cuda.synchronize()
        r.watch.phase(None)
    r.barrier = barrier
    return r


def synthetic_inputs(args, r):
    """The rank's batch: reference seed 1, sites seed 2, pileups seed 3, generated on the device and resident in HBM; the
    site set; the phase-1 SNP records of its samples.  Returns a namespace."""
    import types
    from snp_pipeline_amd import _lib as L
    torch, d = r.torch, r.d
    x = types.SimpleNamespace()
    G, S = args.genome, args.sites
    if args.scaling == "weak":
        x.n_total = r.world * args.samples
        x.g0, x.g1 = r.rank * args.samples, (r.rank + 1) * args.samples
    else:
        x.n_total = args.samples
        x.g0, x.g1 = r.sharding.shard_bounds(x.n_total, r.rank, r.world)
    B = x.B = x.g1 - x.g0                                     # this rank's samples: global indices [g0, g1)
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
        sizes.append(d.synth_pileup_dev(3, x.g0 + i, G, ref.data_ptr(), alt.data_ptr(), 0, 0, mean_depth=args.depth))
    offs = np.zeros(B + 1, dtype=np.uint64)
    for i, n in enumerate(sizes):
        offs[i + 1] = offs[i] + ((n + 255) // 256) * 256
    pile = torch.empty(int(offs[-1]) + 256, dtype=torch.uint8, device="cuda")
    for i in range(B):
        n = d.synth_pileup_dev(3, x.g0 + i, G, ref.data_ptr(), alt.data_ptr(), pile.data_ptr() + int(offs[i]),
                               sizes[i], mean_depth=args.depth)
        assert n == sizes[i]
    torch.cuda.synchronize()
    x.G, x.S, x.ref, x.refh, x.pos, x.alt, x.sizes, x.offs, x.pile = G, S, ref, refh, pos, alt, sizes, offs, pile
    x.pile_bytes = int(sum(sizes))
    x.ss = d.siteset([(b"synth_chr1", int(p)) for p in pos], [L.SITE_IN_SNPLIST] * S)
    x.prm = r.dev.make_params(0, 0.6, 3, 0, 0.0)              # pipeline defaults (snppipeline.conf:249)
    # phase-1 SNP records of every sample (what merge_sites reads from var.flt.vcf): a fixed subset of the sites per
    # sample, keyed by the global sample index so that every rank can also tell what the union must be
    x.recs = min(args.vcf_records, S)
    x.sample_records = lambda g: np.sort(np.random.default_rng(1000 + g).choice(pos, size=x.recs, replace=False))     # noqa: E731
    x.local_keys = torch.from_numpy(np.concatenate([x.sample_records(g) for g in range(x.g0, x.g1)] + [np.zeros(0, np.int64)]).astype(np.int64)).cuda()
    x.local_samp = torch.from_numpy(np.repeat(np.arange(x.g0, x.g1, dtype=np.int32), x.recs)).cuda()
    return x


PHASES = ("c1_gather_and_site_union", "scan_and_call", "pack_and_c2_row_gather", "distance_tiles", "row_band_exchange")


def timed_steps(args, r, x):
    """W warm-up steps, then exactly K steps between barriers; checks of what they left.  Returns a namespace of timings and buffers."""
    import types
    torch, dist, d, sharding, watch = r.torch, r.dist, r.d, r.sharding, r.watch
    B, S, n_total, rank, world = x.B, x.S, x.n_total, r.rank, r.world
    t = types.SimpleNamespace()
    n_records = t.n_records = n_total * x.recs
    u_keys = torch.zeros(max(n_records, 1), dtype=torch.int64, device="cuda")
    u_off = torch.zeros(n_records + 1, dtype=torch.int32, device="cuda")
    u_car = torch.zeros(max(n_records, 1), dtype=torch.int32, device="cuda")
    u_n = torch.zeros(4, dtype=torch.int32, device="cuda")
    bases = t.bases = torch.empty((B, S), dtype=torch.uint8, device="cuda")
    filt = torch.empty((B, S), dtype=torch.uint8, device="cuda")
    status = torch.empty((max(B, 1), 4), dtype=torch.int64, device="cuda")
    row_bytes = d.packed_row_bytes(S)
    packed = torch.empty((B, row_bytes), dtype=torch.uint8, device="cuda")
    bands = sharding.RowBands(n_total, world)
    per = (n_total + world - 1) // world
    packed_pad = torch.zeros((max(bands.n_padded, world * per), row_bytes), dtype=torch.uint8, device="cuda")
    dmat = torch.zeros((bands.n_padded, bands.n_padded), dtype=torch.int32, device="cuda")
    sizes_np = np.asarray(x.sizes, dtype=np.uint64)
    band_holder = [None]
    phase_events = []                                         # per timed step: len(PHASES) + 1 events on the stream the kernels run on

    def step(timed=False):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(PHASES) + 1)] if timed else None

        def mark(k):
            watch.phase(PHASES[k] if k < len(PHASES) else "end of step")
            if timed:
                ev[k].record()
        mark(0)
        # C1: every rank's SNP records -> the same site union on every rank (the snplist)
        keys_all, _ = sharding.all_gather_varlen(x.local_keys)
        samp_all, _ = sharding.all_gather_varlen(x.local_samp)
        d.merge_sites_dev(keys_all.data_ptr(), samp_all.data_ptr(), keys_all.numel(), u_keys.data_ptr(), u_off.data_ptr(),
                          u_car.data_ptr(), u_n.data_ptr())
        mark(1)
        # one scan launch and one call launch for the rank's whole batch; sample i is bytes [offs[i], offs[i] + sizes[i])
        if B:
            d.call_consensus_batch_dev(x.ss, x.pile.data_ptr(), x.offs[:B], x.prm, bases.data_ptr(), filt.data_ptr(), status.data_ptr(),
                                       sizes=sizes_np)
        mark(2)
        if B:
            d.pack_matrix_dev(bases.data_ptr(), B, S, S, packed.data_ptr())
        # C2: RCCL all-gather of the packed rows over xGMI when world > 1 (straight into the padded matrix)
        sharding.all_gather_rows_into(packed, n_total, packed_pad)
        mark(3)
        d.distance_packed_dev(packed_pad.data_ptr(), bands.n_padded, S, dmat.data_ptr(), rank, world)
        mark(4)
        band_holder[0] = bands.exchange(dmat, rank)           # every rank: the complete rows of its band
        mark(5)
        if timed:
            phase_events.append(ev)

    for _ in range(args.warmup):
        step()
    r.barrier()
    d.kernel_timing(True)
    d.kernel_time_ms(0), d.kernel_time_ms(1), d.kernel_time_ms(2)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    r.barrier()
    elapsed = time.perf_counter() - t0
    # where the step's time goes on this rank (events on the kernels' stream; with gloo the collectives are host work between
    # them), and the slowest rank per phase
    phase_ms = [sum(ev[k].elapsed_time(ev[k + 1]) for ev in phase_events) / max(len(phase_events), 1) for k in range(len(PHASES))]
    phase_max = list(phase_ms)
    phase_all = [list(phase_ms)]
    if r.multi:
        pt = torch.tensor(phase_ms, dtype=torch.float64, device="cpu" if r.one_gpu else "cuda")
        every = [torch.zeros_like(pt) for _ in range(world)]
        dist.all_gather(every, pt)
        phase_all = [[float(v) for v in e.tolist()] for e in every]
        phase_max = [max(p[k] for p in phase_all) for k in range(len(PHASES))]
    t.phase_ms, t.phase_max, t.phase_all = phase_ms, phase_max, phase_all
    t.scan_ms, t.scan_n = d.kernel_time_ms(0)
    t.call_ms, _ = d.kernel_time_ms(1)
    t.dist_ms, _ = d.kernel_time_ms(2)
    d.kernel_timing(False)
    if r.multi:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if r.one_gpu else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    t.elapsed = elapsed
    # ---- what the steps left ------------------------------------------------------------------------------------------------------
    if B:
        st = status[:B].cpu().numpy()
        if (st[:, 0] != -1).any():
            raise SystemExit("scan reported a malformed pileup: %r" % st[:, 0])
        if (filt.cpu().numpy() & 0x80).any():
            raise SystemExit("caller reported a malformed line")
    want_union = len(np.unique(np.concatenate([x.sample_records(g) for g in range(n_total)]))) if n_total * x.recs <= 4_000_000 else None
    if want_union is not None and int(u_n[0]) != want_union:
        raise SystemExit("site union has %d keys, expected %d" % (int(u_n[0]), want_union))
    t.unique_sites, t.carriers = int(u_n[0]), int(u_n[1])
    if args.dump:
        lo, hi = bands.band_rows(rank)
        np.savez(args.dump + ".rank%d.npz" % rank, union=u_keys[:t.unique_sites].cpu().numpy(), union_off=u_off[:t.unique_sites + 1].cpu().numpy(),
                 carriers=u_car[:t.carriers].cpu().numpy(), packed=packed_pad[:n_total].cpu().numpy(),
                 band=band_holder[0][:hi - lo, :n_total].cpu().numpy(), band_rows=np.array([lo, hi]), bases=bases.cpu().numpy(),
                 first_sample=np.array([x.g0, x.g1]))
    return t


def headline(args, r, x, t):
    """The result object of the timed steps: the contract's keys, the workload, who made the exchanges, K1's roofline, the phases."""
    dist, world, multi = r.dist, r.world, r.multi
    per_step = t.elapsed / args.steps
    scan_avg_ms = t.scan_ms / max(t.scan_n, 1)
    algo_bytes = x.pile_bytes * args.steps / max(t.scan_n, 1)   # per launch: the rank's whole batch of pileup text
    achieved = algo_bytes / (scan_avg_ms * 1e-3) / 1e9 if scan_avg_ms > 0 else 0.0
    # HBM traffic of one scan launch from the committed PMC passes of this very workload: what stands in where the run cannot measure it
    # itself (N > 1, --no-live-traffic, no rocprofv3)
    pt, pt_src = rows.committed_traffic("pmc_traffic.json", lambda p: (p["workload"]["samples_per_gpu"], p["workload"]["genome_bp"], p["workload"]["mean_depth"],
                                                                       p["workload"]["snp_sites"]) == (x.B, x.G, args.depth, x.S))
    traffic = pt["traffic_bytes_per_launch"] if pt else None
    traffic_note = ("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of this workload; not of this very run)" % pt_src) if pt else None
    return {
        "metric": "consensus_bases_called_per_sec", "value": x.n_total * x.S / per_step, "unit": "bases/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": per_step * 1e3, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3] shard (%d samples%s x %d bp x %gx synthetic pileups, %d SNP sites, %d SNP records "
                               "per sample; the reference bundles no pileups); step = site-record gather + site union, one batched "
                               "scan launch + one call launch, 4-bit pack, row all-gather, all-pairs distance tiles, row-band exchange"
                               % (args.samples, "/GPU" if args.scaling == "weak" else " in total", x.G, args.depth, x.S, x.recs),
                   "samples_total": x.n_total, "samples_this_rank": x.B, "genome_bp": x.G, "mean_depth": args.depth, "snp_sites": x.S,
                   "pileup_bytes_this_rank": x.pile_bytes, "caller": "q0 c0.6 D3 d0 b0",
                   "parallelism": "samples sharded over %d rank(s)%s"
                                  % (world, (", backend %s, world size %d" % (r.backend, dist.get_world_size())) if multi else "")},
        "comm": {"backend": (dist.get_backend() if multi else None), "world_size": (dist.get_world_size() if multi else 1),
                 "launcher": "bench.py started its own ranks (torch.distributed.run, 127.0.0.1)" if os.environ.get("SNPGPU_BENCH_LAUNCHER") == "self"
                 else ("torch.distributed.run around bench.py" if multi else "none (one process)"),
                 "collectives_per_step": "C1 variable-length all-gather of site records, C2 all-gather of packed rows, one all-to-all of distance tiles" if multi else "none"},
        "comm_route": ({"exchanges": "libsnpgpu.so (snpgpu_allgather / snpgpu_allgatherv / snpgpu_alltoallv on the context's stream, csrc/comm.hip)",
                        "rccl_version": r.comm_info["rccl_version"], "world_size_rccl_reports": r.comm_info["rccl_comm_count"], "rank": r.comm_info["rank"]}
                       if r.abi_route else ({"exchanges": "torch.distributed (%s)" % r.backend} if multi else None)),
        "genome_bp_per_sec": x.n_total * x.G / per_step,
        "pileup_gb_per_sec": (x.pile_bytes * x.n_total / max(x.B, 1)) / per_step / 1e9,
        "roofline": {"kernel": "k_scan_wave", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_over_algorithmic": (traffic / algo_bytes) if traffic and algo_bytes else None,
                     "traffic_source": traffic_note, "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": scan_avg_ms, "launches": t.scan_n},
        "kernels_ms_per_step": {"k_scan_wave": t.scan_ms / args.steps, "k_call_sites": t.call_ms / args.steps, "k_distance": t.dist_ms / args.steps},
        "site_union": {"records": t.n_records, "unique_sites": t.unique_sites, "carriers": t.carriers},
        "phases_ms_per_step": {"rank0": dict(zip(PHASES, t.phase_ms)), "max_over_ranks": dict(zip(PHASES, t.phase_max)),
                               "per_rank": [dict(zip(PHASES, p)) for p in t.phase_all],
                               "note": "device time between events on the kernels' stream, averaged over the timed steps"},
    }


def side_rows(args, r, x, t, out):
    """Everything measured after the timed region (bench_rows.py), into `out`; most of it on rank 0 at N = 1 only."""
    from snp_pipeline_amd import _lib as L
    torch, d, dev = r.torch, r.d, r.dev
    B, G, S = x.B, x.G, x.S
    alone = r.rank == 0 and r.world == 1 and B > 0
    # the shard as ONE job from files to files (hot_path_batch); first: the device memory it takes has not been through the other rows'
    # allocations and frees, as in a job of its own
    if alone and args.pipeline_files > 0:
        out["pipeline_from_files"] = rows.side_row(rows.pipeline_from_files, x.pile, x.offs, x.sizes, x.refh, G, min(args.pipeline_files, B), not args.skip_separate_steps)
    # secondary metric: the distance step alone at configs[4] shape (kernel + row-band exchange)
    if not args.skip_secondary:
        out["secondary"] = rows.secondary_distance(d, r.sharding, args, r.rank, r.world, r.multi, r.one_gpu, r.barrier, r.watch)
    # context figures: a measured device-copy ceiling and the small latency-bound steps
    if r.rank == 0 and r.world == 1 and not args.skip_aux:
        out["roofline"]["measured_copy_gbps_read_plus_write"] = rows.device_copy_gbps(torch)
        out["aux_steps_ms"] = rows.aux_steps(d, x.pos, G)
    # the scan kernel on its weak shapes (shallow / deep pileups, CR LF, many contigs)
    if alone and args.shape_samples > 0:
        out["scan_shapes"] = rows.side_row(rows.scan_shapes, d, L, dev, x.ref, x.alt, G, x.pos, args.shape_samples)
    # end to end: pileup FILES in the page cache -> consensus bytes on the host, through the streamed ingestion
    if alone and args.e2e_files > 0:
        out["end_to_end"] = rows.side_row(rows.end_to_end, d, x.ss, x.prm, x.pile, x.offs, x.sizes, t.bases, min(args.e2e_files, B), S)
    # the call paths the headline leaves out: per-site counts (the default configuration's consensus.vcf), the strict caller, --vcfAllPos
    if alone and not args.skip_call_variants:
        out["call_variants"] = rows.side_row(rows.call_variants, d, x.ss, x.prm, x.pile, x.offs, x.sizes, S, dev, torch, x.pos)
    # phase-1 site calling on files
    if alone and args.site_files > 0:
        out["site_calling"] = rows.side_row(rows.site_calling, d, x.pile, x.offs, x.sizes, min(args.site_files, B))
    # CPU baseline: the oracle on samples of the batch
    if alone and args.cpu_samples > 0:
        out["cpu_baseline"] = rows.cpu_baseline(args, d, x.pile, x.offs, x.sizes, t.bases, x.pos, G, S, out["value"], out.get("secondary"))
        rows.from_files_ratios(out, S)
    # the scan's HBM traffic, measured now (two short child runs under rocprofv3 --pmc)
    if alone and not args.no_live_traffic:
        r.watch.phase(None)
        torch.cuda.empty_cache()                                 # (the child generates the same 54 GB beside this process's)
        live = rows.live_traffic(args, out["roofline"]["algorithmic_bytes_per_launch"])
        if "error" in live:
            out["roofline"]["traffic_live_error"] = live["error"]
        else:
            out["roofline"].update(live)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    real_stdout = claim_stdout()
    r = open_rank(args)
    x = synthetic_inputs(args, r)
    t = timed_steps(args, r, x)
    out = headline(args, r, x, t)
    side_rows(args, r, x, t, out)
    line = None
    if r.rank == 0:
        out["north_star"] = rows.north_star(out, args, r.world, x.S)
        line = json.dumps(compact(out), separators=(",", ":"))
        if len(line) >= COMPACT_LIMIT:                          # never lose the headline to its own length again (BENCH_r05: parsed = null)
            line = json.dumps(compact(out, minimal=True), separators=(",", ":"))
        write_detail(args.detail, out)
    if r.multi:
        r.dist.destroy_process_group()
    if r.rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (line + "\n").encode())


def write_detail(path, out):
    """The full result beside the compact line; a box that offers no place for it costs the run nothing."""
    if not path:
        return
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
    except OSError as err:
        sys.stderr.write("bench.py: detail file %s not written: %s\n" % (path, err))


def _r(x, digits=6):
    """Floats of the compact line at 6 significant digits."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def _pick(src, *keys):
    return {k: src[k] for k in keys if isinstance(src, dict) and k in src}


def compact(out, minimal=False):
    """The line the driver parses: the contract's keys, `roofline` and `cpu_baseline`, and one number each from the side rows.  No
    prose, no tables: those live in the detail file."""
    c = _pick(out, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    c["config"] = _pick(out["config"], "workload", "samples_total", "samples_this_rank", "genome_bp", "mean_depth", "snp_sites", "pileup_bytes_this_rank", "parallelism")
    c["comm"] = _pick(out["comm"], "backend", "world_size")
    cr = out.get("comm_route")
    c["comm_route"] = None if not cr else ("libsnpgpu.so over rccl %s" % cr["rccl_version"] if "rccl_version" in cr else cr["exchanges"])
    c["roofline"] = _pick(out["roofline"], "kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic",
                          "algorithmic_bytes_per_launch", "avg_launch_ms", "launches", "measured_copy_gbps_read_plus_write")
    src = out["roofline"].get("traffic_source") or ""
    c["roofline"]["traffic_is"] = "measured in this run" if src.startswith("measured in this run") else ("committed profile of this workload" if src else None)
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = _pick(cb, "value", "unit", "cores", "kind", "sample", "matches_gpu", "gpu_over_cpu_1core")
        if "parallel" in cb:
            c["cpu_baseline"]["parallel"] = _pick(cb["parallel"], "value", "processes", "usable_cores", "matches_gpu")
        if "distance" in cb:
            c["cpu_baseline"]["distance"] = _pick(cb["distance"], "value", "unit", "site_compares_per_sec", "matches_gpu")
    if minimal:
        c["config"]["workload"] = c["config"]["workload"][:300]
        c["cpu_baseline"] = _pick(c.get("cpu_baseline") or {}, "value", "unit", "cores", "kind", "matches_gpu")
        return _r(c)
    if "secondary" in out:
        c["secondary"] = _pick(out["secondary"], "metric", "value", "unit", "site_compares_per_sec", "valu_frac_of_peak", "kernel_ms")
    c["kernels_ms_per_step"] = out.get("kernels_ms_per_step")
    c["phases_ms_max_over_ranks"] = out["phases_ms_per_step"]["max_over_ranks"]
    cv = out.get("call_variants") or {}
    for key, name in (("strict", "k2_frac"), ("call_with_counts", "k2_with_counts_frac")):
        if isinstance(cv.get(key), dict) and "roofline" in cv[key]:
            c[name] = cv[key]["roofline"]["frac"]
    if isinstance(cv.get("all_positions"), dict) and "ms_per_step" in cv["all_positions"]:
        c["all_positions_ms"] = cv["all_positions"]["ms_per_step"]
        c["all_positions_file_to_vcf_ms"] = (cv["all_positions"].get("file_to_vcf_file") or {}).get("ms")
    sc = out.get("site_calling") or {}
    if "roofline" in sc:
        c["site_calling_frac"] = sc["roofline"]["frac"]                # one sample per launch: what the pipeline runs
        c["site_calling_batch_frac"] = (sc["roofline"].get("batch_over_the_shard_not_used_by_the_pipeline") or {}).get("frac")
    shapes = out.get("scan_shapes") or {}
    if shapes and "error" not in shapes:
        c["scan_shapes_frac"] = {k: v["frac_of_hbm_peak"] for k, v in shapes.items() if isinstance(v, dict) and "frac_of_hbm_peak" in v}
    pipe = out.get("pipeline_from_files") or {}
    if "samples_per_sec" in pipe:
        c["pipeline_from_files"] = _pick(pipe, "samples", "seconds", "samples_per_sec", "pileup_gb_per_sec")
    e2e = out.get("end_to_end") or {}
    if "pileup_gb_per_sec" in e2e:
        c["end_to_end"] = _pick(e2e, "files", "pileup_gb_per_sec", "pinned_h2d_gb_per_sec")
    ns = out.get("north_star") or {}
    c["north_star"] = _pick(ns, "hbm_target_met", "ratio", "ratio_consensus_only", "ratio_distance_only", "ratio_target_met")
    errors = [k for k, v in out.items() if isinstance(v, dict) and "error" in v]
    if errors:
        c["side_rows_with_errors"] = errors
    c["detail"] = "gpurun_out/bench_detail.json"
    return _r(c)


if __name__ == "__main__":
    main()

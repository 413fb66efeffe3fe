#!/usr/bin/env python3
"""Development helper: sweep the streamed-ingestion knobs (chunk size, readers, staging buffers, slots) over pileup files
in the page cache.  Usage: python tools/stream_tune.py [n_files] [genome_len] [own|torch]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from snp_pipeline_amd import _lib as L
    from snp_pipeline_amd import device as dev
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
    mode = sys.argv[3] if len(sys.argv) > 3 else "own"
    S = G // 100
    d = dev.Device(0)
    d.use_torch_stream()
    ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda")
    d.synth_reference_dev(1, G, ref.data_ptr())
    pos = np.sort(np.random.default_rng(2).choice(np.arange(501, G - 499), size=S, replace=False))
    alt_h = np.zeros(G + 1, dtype=np.uint8)
    alt_h[pos] = ord("A")
    alt = torch.from_numpy(alt_h).cuda()
    tmpdir = tempfile.mkdtemp(prefix="snptune_", dir=os.environ.get("SNPTUNE_DIR", tempfile.gettempdir()))
    paths, total = [], 0
    try:
        for i in range(B):
            n = d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), 0, 0)
            buf = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
            d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), buf.data_ptr(), n + 16)
            torch.cuda.synchronize()
            path = os.path.join(tmpdir, "s%d.pileup" % i)
            with open(path, "wb") as f:
                f.write(buf[:n].cpu().numpy().tobytes())
            paths.append(path)
            total += n
        ss = d.siteset([(b"synth_chr1", int(p)) for p in pos], [L.SITE_IN_SNPLIST] * S)
        prm = dev.make_params(0, 0.6, 3, 0, 0.0)
        if mode == "own":
            d._check(d.lib.snpgpu_ctx_reset_stream(d.ctx))
        d.call_consensus_files(ss, paths[:1], prm)
        print("%d files, %.2f GB, stream=%s" % (B, total / 1e9, mode))
        for chunk in (8 << 20, 16 << 20, 32 << 20):
            for readers, staging in ((4, 8), (8, 12), (16, 24)):
                for slots in (2, 3):
                    d.call_consensus_files(ss, paths[:2], prm, chunk_bytes=chunk, n_readers=readers, n_staging=staging, n_slots=slots)
                    t0 = time.perf_counter()
                    _, rcs, st = d.call_consensus_files(ss, paths, prm, chunk_bytes=chunk, n_readers=readers, n_staging=staging, n_slots=slots)
                    dt = time.perf_counter() - t0
                    print("chunk %2d MiB readers %2d staging %2d slots %d: %6.1f GB/s  wall %.3f s  wait_read %.3f wait_dev %.3f enqueue %.3f | readers: reading %.3f waiting %.3f"
                          % (chunk >> 20, readers, staging, slots, total / dt / 1e9, dt, st.seconds_waiting_for_readers, st.seconds_waiting_for_device,
                             st.seconds_enqueueing, st.reader_seconds_reading, st.reader_seconds_waiting), flush=True)
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)


if __name__ == "__main__":
    main()

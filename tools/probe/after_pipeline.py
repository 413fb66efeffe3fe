"""After an in-process hot_path_batch: does the next torch allocation fail?  (AMD_LOG_LEVEL=3 shows which HIP call returned what.)"""
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from snp_pipeline_amd import device as dev  # noqa: E402

d = dev.Device(0)
d.use_torch_stream()
G = 200_000
ref = torch.empty(G + 1, dtype=torch.uint8, device="cuda")
d.synth_reference_dev(1, G, ref.data_ptr())
torch.cuda.synchronize()
refh = ref.cpu().numpy()
pos = np.unique(np.random.default_rng(2).choice(np.arange(501, G - 499, dtype=np.int64), size=2000, replace=False))
alt_h = np.zeros(G + 1, dtype=np.uint8)
alt_h[pos] = ord("A")
alt = torch.from_numpy(alt_h).cuda()


def sample_bytes(i):
    size = d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), 0, 0)
    buf = torch.empty(size + 256, dtype=torch.uint8, device="cuda")
    d.synth_pileup_dev(3, i, G, ref.data_ptr(), alt.data_ptr(), buf.data_ptr(), size)
    torch.cuda.synchronize()
    return buf[:size].cpu().numpy()


tmpdir, ref_path, dirs_file, dirs, total = bench.write_sample_tree(tempfile.gettempdir(), refh, G, sample_bytes, 4)
try:
    sys.stderr.write("=== BEFORE hot_path_batch\n")
    bench.run_cli(bench.hot_path_line(dirs_file, ref_path))
    sys.stderr.write("=== AFTER hot_path_batch\n")
    x = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    print("torch.empty after the job: ok", x.numel())
finally:
    shutil.rmtree(tmpdir, ignore_errors=True)

python -c "import torch" 2>/dev/null
for v in lo lo2; do
export SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_$v.so
echo "== $v 125 x 30x"; SWEEP_REPS=3 timeout 300 python tools/scan_sweep.py 125 30 "" 2>&1 | grep -v amdgpu.ids
echo "== $v 125 x 8x"; SWEEP_REPS=3 timeout 300 python tools/scan_sweep.py 125 8 ""  2>&1 | grep -v amdgpu.ids
echo "== $v 64 x 15x"; SWEEP_REPS=3 timeout 300 python tools/scan_sweep.py 64 15 "" 2>&1 | grep -v amdgpu.ids
echo "== $v 16 x 100x"; SWEEP_REPS=3 timeout 300 python tools/scan_sweep.py 16 100 ""  2>&1 | grep -v amdgpu.ids
echo "== $v 48 x 30x"; SWEEP_REPS=3 timeout 300 python tools/scan_sweep.py 48 30 ""  2>&1 | grep -v amdgpu.ids
done
timeout 600 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -3

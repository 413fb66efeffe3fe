python -c "import torch" 2>/dev/null
timeout 400 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -3
for v in lo2 lo4 lo3 lo2 lo4; do
export SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_$v.so
echo "== $v crlf"; timeout 200 python tools/scan_crlf.py 2>&1 | tail -3 | head -2
echo "== $v 125 x 30x"; SWEEP_REPS=3 timeout 300 python tools/scan_sweep.py 125 30 "" 2>&1 | grep -v amdgpu.ids
done

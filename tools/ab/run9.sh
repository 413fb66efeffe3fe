python -c "import torch" 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -3
for v in lo4 lo5 lo4 lo5; do
export SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_$v.so
echo "== $v 125 x 30x"; SWEEP_REPS=2 timeout 300 python tools/scan_sweep.py 125 30 "" 2>&1 | grep "GB/s"
echo "== $v 125 x 8x"; SWEEP_REPS=2 timeout 300 python tools/scan_sweep.py 125 8 "" 2>&1 | grep "GB/s"
echo "== $v 100 x 15x"; SWEEP_REPS=2 timeout 300 python tools/scan_sweep.py 100 15 "" 2>&1 | grep "GB/s"
done
unset SNPGPU_TUNE_LIB
timeout 420 python tools/fuzz_campaign.py 360 > gpurun_out/r4c_fuzz_campaign.log 2>&1; tail -12 gpurun_out/r4c_fuzz_campaign.log

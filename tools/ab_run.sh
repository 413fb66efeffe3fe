mkdir -p gpurun_out
python -m pytest tests/test_gpu_consensus.py tests/test_gpu_stream.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
for depth in 30 15 100; do
  n=48; [ $depth = 15 ] && n=80; [ $depth = 100 ] && n=16
  for rep in 1 2 3; do
    echo "== depth $depth A (HEAD)"; SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_a.so python tools/scan_tune.py $n 5000000 batch $depth 2>/dev/null | tail -1
    echo "== depth $depth B (tree)"; python tools/scan_tune.py $n 5000000 batch $depth 2>/dev/null | tail -1
  done
done

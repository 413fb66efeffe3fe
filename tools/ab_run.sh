mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for depth in 30 100 200; do
  echo "== counts path depth $depth A (HEAD)"; SNPGPU_TUNE_COUNTS=1 SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_a.so python tools/scan_tune.py 8 5000000 single $depth 2>/dev/null | tail -1
  echo "== counts path depth $depth B (tree)"; SNPGPU_TUNE_COUNTS=1 python tools/scan_tune.py 8 5000000 single $depth 2>/dev/null | tail -1
done

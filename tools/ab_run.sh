mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for depth in 30 15 60 100 200; do
  n=48; [ $depth = 15 ] && n=80; [ $depth = 60 ] && n=24; [ $depth = 100 ] && n=16; [ $depth = 200 ] && n=8
  echo "== depth $depth A (HEAD)"; SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_a.so python tools/scan_tune.py $n 5000000 batch $depth 2>/dev/null | tail -1
  echo "== depth $depth B (tree)"; python tools/scan_tune.py $n 5000000 batch $depth 2>/dev/null | tail -1
done
for depth in 30 100; do
  echo "== counts path depth $depth A (HEAD)"; SNPGPU_TUNE_COUNTS=1 SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_a.so python tools/scan_tune.py 8 5000000 single $depth 2>/dev/null | tail -1
  echo "== counts path depth $depth B (tree)"; SNPGPU_TUNE_COUNTS=1 python tools/scan_tune.py 8 5000000 single $depth 2>/dev/null | tail -1
done

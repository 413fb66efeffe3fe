mkdir -p gpurun_out
python -m pytest tests/test_gpu_consensus.py tests/test_gpu_stream.py tests/test_gpu_cli.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -5
for depth in 30 15 100 8; do
  n=48; [ $depth = 15 ] && n=80; [ $depth = 100 ] && n=16; [ $depth = 8 ] && n=120
  for rep in 1 2; do
    echo "== depth $depth A (HEAD)"; SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_a.so python tools/scan_tune.py $n 5000000 batch $depth 2>/dev/null | tail -1
    echo "== depth $depth B (tree)"; python tools/scan_tune.py $n 5000000 batch $depth 2>/dev/null | tail -1
  done
done
echo "== CRLF A (HEAD)"; SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_a.so python tools/scan_crlf.py 2>/dev/null | tail -3
echo "== CRLF B (tree)"; python tools/scan_crlf.py 2>/dev/null | tail -3
for cfg in "40 125000" "400 12500"; do
  echo "== multi contig $cfg A (HEAD)"; SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_a.so python tools/scan_multi_contig.py $cfg 2>/dev/null | tail -1
  echo "== multi contig $cfg B (tree)"; python tools/scan_multi_contig.py $cfg 2>/dev/null | tail -1
done

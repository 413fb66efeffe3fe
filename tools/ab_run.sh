mkdir -p gpurun_out
python -m pytest tests/test_gpu_consensus.py tests/test_gpu_stream.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -3
echo "== CRLF A (HEAD)"; SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_a.so python tools/scan_crlf.py 2>/dev/null | tail -3
echo "== CRLF B (tree)"; python tools/scan_crlf.py 2>/dev/null | tail -3
for cfg in "40 125000" "400 12500"; do
  echo "== multi contig $cfg A (HEAD)"; SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_a.so python tools/scan_multi_contig.py $cfg 2>/dev/null | tail -2
  echo "== multi contig $cfg B (tree)"; python tools/scan_multi_contig.py $cfg 2>/dev/null | tail -2
done
for depth in 30 15; do
  n=48; [ $depth = 15 ] && n=80
  echo "== depth $depth A (HEAD)"; SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_a.so python tools/scan_tune.py $n 5000000 batch $depth 2>/dev/null | tail -1
  echo "== depth $depth B (tree)"; python tools/scan_tune.py $n 5000000 batch $depth 2>/dev/null | tail -1
done

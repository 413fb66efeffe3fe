# usage: sh tools/ab_run.sh v1 v2 ...   (names of tools/ab/libsnpgpu_<name>.so)
for depth in ${DEPTHS:-30 100 15 8}; do
  n=48; [ $depth = 15 ] && n=64; [ $depth = 8 ] && n=64; [ $depth = 100 ] && n=16
  for rep in 1 2; do
    for v in "$@"; do
      echo "== depth $depth $v: $(SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_$v.so timeout 90 python tools/scan_tune.py $n 5000000 batch $depth 2>&1 | tail -1 | cut -c28-75,150-230)"
    done
  done
done

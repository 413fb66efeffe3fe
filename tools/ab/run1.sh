run() { # name env... 
  v=$1; shift
  for dp in 30 8; do n=48; [ $dp = 8 ] && n=64
    for rep in 1 2; do echo "== $v $* depth $dp: $(env "$@" SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_$v.so timeout 90 python tools/scan_tune.py $n 5000000 batch $dp 2>&1 | tail -1 | cut -c28-75)"; done
  done
}
run lo A=1
run lo SNPGPU_SCAN_WAVES=12
run nb3 A=1
run nb3 SNPGPU_SCAN_WAVES=8
run nb4 SNPGPU_SCAN_WAVES=8
run lo SNPGPU_SCAN_SHARE=250,250,250,250
run lo SNPGPU_SCAN_SHARE=300,270,235,195
run lo SNPGPU_SCAN_SHARE=360,290,205,145

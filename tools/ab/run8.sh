python -c "import torch" 2>/dev/null
export SNPGPU_TUNE_LIB=$PWD/tools/ab/libsnpgpu_lo4.so
rocm-smi --showclocks 2>/dev/null | grep -i "clk" | head -8
for i in 1 2 3; do
echo "== process $i"; SWEEP_REPS=500 timeout 600 python tools/scan_sweep.py 125 30 "" 2>&1 | grep "GB/s" | awk 'NR%25==1'
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "clk\|power\|temp" | head -12
done

python -c "import torch" 2>/dev/null
export SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_lo.so
echo "== 125 x 30x"; timeout 300 python tools/scan_sweep.py 125 30 "" "WAVES=12" "WAVES=12 OVERSUB=1" "WAVES=12 OVERSUB=4" "OVERSUB=1" "OVERSUB=4" "WAVES=8" "SHARE=300,270,235,195" "SHARE=345,288,215,152" "SHARE=315,277,230,178" "WAVES=12 SHARE=300,270,235,195" "WAVES=12 SHARE=345,288,215,152" 2>&1 | tail -30
echo "== 64 x 8x"; timeout 200 python tools/scan_sweep.py 64 8 "" "WAVES=12" "SHARE=300,270,235,195" "SHARE=280,262,240,215" "SHARE=315,277,230,178" "OVERSUB=1" 2>&1 | tail -14
echo "== 64 x 15x"; timeout 200 python tools/scan_sweep.py 64 15 "" "WAVES=12" "SHARE=300,270,235,195" "SHARE=315,277,230,178" 2>&1 | tail -10

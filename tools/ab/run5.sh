python -c "import torch" 2>/dev/null
export SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_lo2.so
A=329,282,223,169
echo "== 125 x 8x"; timeout 300 python tools/scan_sweep.py 125 8 "" "SHARE_DENSE=$A" "OVERSUB=2" "OVERSUB=2 SHARE_DENSE=$A" "OVERSUB=8" "SHARE_DENSE=265,255,245,235" "SHARE_DENSE=290,266,238,206" 2>&1 | grep -v amdgpu.ids
echo "== 100 x 15x"; timeout 300 python tools/scan_sweep.py 100 15 "" "SHARE_DENSE=$A" "OVERSUB=2" "OVERSUB=8" "SHARE_DENSE=270,258,243,225"  2>&1 | grep -v amdgpu.ids

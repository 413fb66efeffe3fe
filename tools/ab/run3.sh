python -c "import torch" 2>/dev/null
export SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_lo.so
A=329,282,223,169; B=315,277,230,178; C=300,270,235,195; D=280,262,240,215; E=265,255,245,235; F=250,250,250,250; G=345,288,215,152
echo "== 125 x 30x"; timeout 300 python tools/scan_sweep.py 125 30 "OVERSUB=2" "OVERSUB=4" "OVERSUB=8" "OVERSUB=16" "OVERSUB=4 SHARE=$B" "OVERSUB=4 SHARE=$G" "OVERSUB=4 SHARE=$C" "OVERSUB=8 SHARE=$C" "OVERSUB=8 SHARE=$F" 2>&1 | grep -v amdgpu.ids
echo "== 125 x 8x"; timeout 300 python tools/scan_sweep.py 125 8 "OVERSUB=2" "OVERSUB=2 SHARE=$C" "OVERSUB=2 SHARE=$D" "OVERSUB=2 SHARE=$E" "OVERSUB=2 SHARE=$F" "OVERSUB=4 SHARE=$D" "OVERSUB=4 SHARE=$E" "OVERSUB=8 SHARE=$E" "OVERSUB=8 SHARE=$F" "OVERSUB=1 SHARE=$D" 2>&1 | grep -v amdgpu.ids
echo "== 125 x 15x"; timeout 300 python tools/scan_sweep.py 125 15 "OVERSUB=2" "OVERSUB=2 SHARE=$B" "OVERSUB=2 SHARE=$C" "OVERSUB=2 SHARE=$D" "OVERSUB=4 SHARE=$C" "OVERSUB=4 SHARE=$D" "OVERSUB=8 SHARE=$D" 2>&1 | grep -v amdgpu.ids
echo "== 40 x 100x"; timeout 300 python tools/scan_sweep.py 40 100 "OVERSUB=2" "OVERSUB=4" "OVERSUB=8" "OVERSUB=4 SHARE=$G" "OVERSUB=4 SHARE=$B" "OVERSUB=4 SHARE=360,295,205,140" 2>&1 | grep -v amdgpu.ids

#!/bin/bash
# Development helper (tuning build tools/ab/libsnpgpu_t.so): site calling on one resident sample / a batch with 2, 4, 6, 8 waves per CU
# and several grid multipliers.  Usage (GPU box): tools/vs_occupancy.sh [depth] [batch]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
dp=${1:-30}; nb=${2:-8}
for w in 8 6 4 2; do
    echo "== $w waves per CU"
    SNPGPU_LIB=$root/tools/ab/libsnpgpu_t.so SNPGPU_VS_WG_WAVES=$w python $root/tools/varscan_kernel_time.py 5000000 $dp 10 $nb 2>&1 | grep -v amdgpu.ids
done
for m in 1 2 4 8; do
    echo "== grid multiplier $m"
    SNPGPU_LIB=$root/tools/ab/libsnpgpu_t.so SNPGPU_VS_GRID_MUL=$m python $root/tools/varscan_kernel_time.py 5000000 $dp 10 $nb 2>&1 | grep -v amdgpu.ids
done
for s in "100,100,100,100" "108,92,92,92" "115,85,85,85" "125,75,75,75"; do
    echo "== share $s"
    SNPGPU_LIB=$root/tools/ab/libsnpgpu_t.so SNPGPU_VS_SHARE=$s python $root/tools/varscan_kernel_time.py 5000000 $dp 10 $nb 2>&1 | grep -v amdgpu.ids
done

#!/bin/bash
# Development helper (tuning build tools/ab/libsnpgpu_t.so, sh tools/variant_build.sh t): site calling over resident samples with several
# age-rank shares of the scan's waves.  Usage (GPU box): tools/vs_share_sweep.sh [depth] [batch] [shares...]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
dp=${1:-30}; nb=${2:-8}; shift; shift
for s in ${@:-"118,100,84,84" "125,100,78,78" "132,100,72,72" "140,100,66,66" "125,105,75,75" "135,95,75,75"}; do
    echo "== share $s"
    SNPGPU_LIB=$root/tools/ab/libsnpgpu_t.so SNPGPU_VS_SHARE=$s python $root/tools/varscan_kernel_time.py 5000000 $dp 10 $nb 2>&1 | grep -v amdgpu.ids
done

#!/bin/bash
# Development helper (GPU box): k_scan_wave with grids larger than what is resident (SNPGPU_SCAN_OVERSUB), smaller workgroups
# (SNPGPU_SCAN_WAVES x SNPGPU_SCAN_BLOCKS_PER_CU) — -DSNPGPU_TUNING build.  Usage: tools/scan_oversub.sh [samples] [depth]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root && SNPGPU_TUNING=1 python -m snp_pipeline_amd.build --force > /dev/null 2>&1
n=${1:-64}; dp=${2:-30}
for cfg in ${SCAN_CFGS:-"16:1:1" "16:1:2" "16:1:4" "16:1:8" "8:2:1" "8:2:4" "4:4:1" "4:4:4" "4:4:8" "2:8:8" "16:1:1"}; do
    IFS=: read w b o <<< "$cfg"
    for rep in 1 2; do
        echo "waves $w blocks/CU $b oversub $o: $(SNPGPU_SCAN_WAVES=$w SNPGPU_SCAN_BLOCKS_PER_CU=$b SNPGPU_SCAN_OVERSUB=$o timeout 120 python tools/scan_tune.py $n 5000000 batch $dp 2>&1 | tail -1 | cut -c28-90,150-200)"
    done
done
cd $root && python -m snp_pipeline_amd.build --force > /dev/null 2>&1

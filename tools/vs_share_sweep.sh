#!/bin/bash
# Development helper (GPU box): k_varscan_scan with different age shares (SNPGPU_VS_SHARE=oldest,..,youngest) — -DSNPGPU_TUNING build.
# Usage: tools/vs_share_sweep.sh [depth] ; shares from $VS_SHARES
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root && SNPGPU_TUNING=1 python -m snp_pipeline_amd.build --force > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
dp=${1:-30}
for sh in ${VS_SHARES:-"100,100,100,100" "110,100,90,80" "120,100,82,70" "130,100,75,60" "105,100,95,90"}; do
    export SNPGPU_VS_SHARE=$sh
    rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/vsw -- python $root/tools/varscan_kernel_time.py 5000000 $dp 6 > /dev/null 2>&1
    f=$(find $root/gpurun_out/vsw -name "*kernel_stats.csv" | head -1)
    echo "depth $dp shares $sh: $(python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'varscan_scan' in r['Name']: print('%.1f us avg, %.1f us min' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))")"
    rm -rf $root/gpurun_out/vsw
done
cd $root && python -m snp_pipeline_amd.build --force > /dev/null 2>&1

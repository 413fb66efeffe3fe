#!/bin/bash
# Development helper (GPU box): k_varscan_scan at several grid shapes (waves per CU, workgroups per resident slot) in a
# -DSNPGPU_TUNING build; prints the kernel's average time per shape.  Usage: tools/vs_sweep.sh [depth]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root && SNPGPU_TUNING=1 python -m snp_pipeline_amd.build --force > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
dp=${1:-30}
for shape in ${VS_SHAPES:-"0 1" "0 2" "0 4" "0 8" "8 1" "8 2" "6 2" "4 4"}; do
    set -- ${shape/:/ }
    export SNPGPU_VS_WAVES=$1 SNPGPU_VS_GRID_MUL=$2
    rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/vsw -- python $root/tools/varscan_kernel_time.py 5000000 $dp 6 > /dev/null 2>&1
    f=$(find $root/gpurun_out/vsw -name "*kernel_stats.csv" | head -1)
    echo "depth $dp waves/CU $1 (0 = default) x grid mul $2: $(python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'varscan_scan' in r['Name']: print('%.1f us avg, %.1f us min' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))")"
    rm -rf $root/gpurun_out/vsw
done
cd $root && python -m snp_pipeline_amd.build --force > /dev/null 2>&1

#!/bin/bash
# rocprofv3 kernel statistics of phase-1 site calling on one resident synthetic sample, at several depths.
# Usage (GPU box): tools/varscan_profile.sh [depths...] > gpurun_out/varscan_profile.log ; per-depth CSVs land in gpurun_out/vs_stats_<depth>.csv
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for dp in ${@:-30 100 8}; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/vs$dp -- python $root/tools/varscan_kernel_time.py 5000000 $dp 10 2>&1 | grep "per call"
    find $root/gpurun_out/vs$dp -name "*kernel_stats.csv" -exec cp {} $root/gpurun_out/vs_stats_$dp.csv \;
    rm -rf $root/gpurun_out/vs$dp
    python $root/tools/varscan_kernel_time.py 5000000 $dp 6 12 2>&1 | grep "^batch"     # (no profiler attached: HIP events, twelve samples per launch)
    echo "== depth $dp, one sample per launch (rocprofv3 --kernel-trace --stats)"
    python - <<PY
import csv
for r in csv.reader(open('$root/gpurun_out/vs_stats_$dp.csv')):
    if r[0] == 'Name' or not any(k in r[0] for k in ('varscan', 'lines_index')):
        continue
    n = r[0].split('::')[-1].split('(')[0] if 'anonymous' in r[0] else r[0].split('(')[0]
    print("  %-26s calls %3s  avg %9.1f us  min %9.1f us" % (n[:26], r[1], float(r[3]) / 1e3, float(r[5]) / 1e3))
PY
done
if [ -n "$VS_PMC" ]; then
    for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
        rocprofv3 --pmc $set --output-format csv -d $root/gpurun_out/vs_pmc -- python $root/tools/varscan_kernel_time.py 5000000 ${VS_PMC} 3 > /dev/null 2>&1
        python $root/tools/pmc_summary.py $root/gpurun_out/vs_pmc | grep -A12 -E "k_varscan"
        rm -rf $root/gpurun_out/vs_pmc
    done
fi

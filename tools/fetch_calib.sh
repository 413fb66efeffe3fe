#!/bin/sh
# FETCH_SIZE / WRITE_SIZE on known byte counts (tools/probe/fetch_calib.hip): the program's own byte counts, then the counters per
# kernel from two separate --pmc passes.  Usage (GPU box): sh tools/fetch_calib.sh [GiB] > gpurun_out/fetch_calib.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
gib=${1:-8}
cd /tmp && export TMPDIR=/tmp
[ -x "$root/tools/probe/fetch_calib" ] || hipcc --offload-arch=gfx950 -O2 -o "$root/tools/probe/fetch_calib" "$root/tools/probe/fetch_calib.hip"
"$root/tools/probe/fetch_calib" "$gib"
rm -rf /tmp/fc_fetch /tmp/fc_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fc_fetch -- "$root/tools/probe/fetch_calib" "$gib" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/fc_write -- "$root/tools/probe/fetch_calib" "$gib" > /dev/null 2>&1
echo "--- FETCH_SIZE (KB per dispatch; two dispatches per kernel)"
python "$root/tools/pmc_summary.py" /tmp/fc_fetch
echo "--- WRITE_SIZE (KB per dispatch)"
python "$root/tools/pmc_summary.py" /tmp/fc_write

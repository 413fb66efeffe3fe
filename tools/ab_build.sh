#!/bin/sh
# Development helper: build tools/ab/libsnpgpu_a.so from the csrc of a git revision (default HEAD), to time it against
# the working tree in one GPU session:  SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_a.so python tools/scan_tune.py ...
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/snp_pipeline_amd/csrc" "$tmp/include" "$root/tools/ab"
git -C "$root" archive "$rev" snp_pipeline_amd/csrc include | tar -x -C "$tmp"
objs=""
for f in "$tmp"/snp_pipeline_amd/csrc/*.hip; do
    o="$tmp/$(basename "$f" .hip).o"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -DSNPGPU_TUNING -c "$f" -o "$o" &
    objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/ab/libsnpgpu_a.so" $objs
rm -rf "$tmp"
echo "built tools/ab/libsnpgpu_a.so from $rev"

#!/bin/sh
# Development helper: build tools/ab/libsnpgpu_<name>.so from the WORKING TREE with extra compiler flags, to time variants of a
# kernel against each other in one GPU session:  sh tools/variant_build.sh pf1 -DSCAN_PREFETCH=1
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$root/tools/ab"
objs=""
for f in "$root"/snp_pipeline_amd/csrc/*.hip; do
    o="$tmp/$(basename "$f" .hip).o"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result -DSNPGPU_TUNING \
        -I"$root/include" "$@" -c "$f" -o "$o" &
    objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/ab/libsnpgpu_$name.so" $objs
rm -rf "$tmp"
echo "built tools/ab/libsnpgpu_$name.so with $*"

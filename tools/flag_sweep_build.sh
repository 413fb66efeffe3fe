#!/bin/sh
# Development helper: libsnpgpu variants whose scan.hip is compiled with different scheduler flags -> tools/ab/lib_<tag>.so
# (time them with SNPGPU_TUNE_LIB=tools/ab/lib_<tag>.so python tools/scan_tune.py ...).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/snp_pipeline_amd/csrc
tmp=$(mktemp -d)
mkdir -p "$root/tools/ab"
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result -I$root/include"
others=""
for f in ctx consensus stream varscan varscan_rows vcf_rows tsv_out fasta_in vcf_in distance regions synth; do
    /opt/rocm/bin/hipcc $base -c $csrc/$f.hip -o $tmp/$f.o &
    others="$others $tmp/$f.o"
done
wait
i=0
while IFS='|' read -r tag flags; do
    [ -z "$tag" ] && continue
    /opt/rocm/bin/hipcc $base $flags -c $csrc/scan.hip -o $tmp/scan_$tag.o 2> $tmp/err_$tag.txt || { echo "$tag: compile failed"; tail -3 $tmp/err_$tag.txt; continue; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/ab/lib_$tag.so $tmp/scan_$tag.o $others -lpthread
    echo "built lib_$tag.so ($flags)"
done <<LIST
base|
maxilp|-mllvm -amdgpu-sched-strategy=max-ilp
LIST
rm -rf "$tmp"

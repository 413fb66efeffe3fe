#!/bin/sh
# Development helper: libsnpgpu variants whose scan tile loads carry a cache-policy modifier -> tools/ab/lib_<tag>.so
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/snp_pipeline_amd/csrc
tmp=$(mktemp -d)
mkdir -p "$root/tools/ab"
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result -I$root/include -I$csrc"
others=""
for f in ctx consensus stream varscan varscan_rows vcf_rows tsv_out fasta_in vcf_in distance regions synth; do
    /opt/rocm/bin/hipcc $base -c $csrc/$f.hip -o $tmp/$f.o &
    others="$others $tmp/$f.o"
done
wait
for tag in plain nt sc1 sc0sc1 sc0sc1nt; do
    case $tag in plain) mod="";; nt) mod=" nt";; sc1) mod=" sc1";; sc0sc1) mod=" sc0 sc1";; sc0sc1nt) mod=" sc0 sc1 nt";; esac
    sed "s/global_load_lds_dwordx4 %0, %1 offset:%3/global_load_lds_dwordx4 %0, %1 offset:%3$mod/" $csrc/scan.hip > $tmp/scan_$tag.hip
    /opt/rocm/bin/hipcc $base -c $tmp/scan_$tag.hip -o $tmp/scan_$tag.o 2> $tmp/err.txt || { echo "$tag: compile failed"; tail -2 $tmp/err.txt; continue; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/ab/lib_$tag.so $tmp/scan_$tag.o $others -lpthread
    echo "built lib_$tag.so"
done
rm -rf "$tmp"

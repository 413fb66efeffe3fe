#!/bin/sh
# Development helper: time the variants of tools/nt_sweep_build.sh (30x, 64 samples, batch mode; two rounds).
for rep in 1 2; do
  for tag in plain nt sc1 sc0sc1 sc0sc1nt; do
    printf "%-9s " $tag
    SNPGPU_TUNE_LIB=tools/ab/lib_$tag.so python tools/scan_tune.py 64 5000000 batch 30 2>/dev/null | tail -1 | cut -c1-150
  done
done

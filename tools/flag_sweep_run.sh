#!/bin/sh
# Development helper: time the variants of tools/flag_sweep_build.sh (30x, 64 samples, batch mode; alternating rounds).
for rep in 1 2 3 4 5; do
  for tag in base maxilp; do
    printf "%-9s " $tag
    SNPGPU_TUNE_LIB=tools/ab/lib_$tag.so python tools/scan_tune.py 64 5000000 batch 30 2>/dev/null | tail -1 | cut -c1-80
  done
done

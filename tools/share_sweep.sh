#!/bin/sh
# Development helper (tuning build only): sweep the per-age-rank tile shares of k_scan_wave.
# Usage: SNPGPU_TUNING=1 build, then  sh tools/share_sweep.sh [n_samples] [depth]
n=${1:-64}; depth=${2:-30}
for sh in 329,282,223,169 250,250,250,250 300,270,235,195 315,277,230,178 345,288,215,152 360,290,205,145 329,282,223,140 329,300,223,169 310,282,240,169; do
  for rep in 1 2; do
    printf "share %s : " $sh
    SNPGPU_SCAN_SHARE=$sh python tools/scan_tune.py $n 5000000 batch $depth 2>/dev/null | tail -1 | cut -c1-90
  done
done

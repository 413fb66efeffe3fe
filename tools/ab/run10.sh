python -c "import torch" 2>/dev/null
export SNPGPU_TUNE_LIB=tools/ab/libsnpgpu_lo5.so
for i in 1 2; do echo "== process $i (125 allocations)"; timeout 300 python tools/scan_realloc.py 125 30 6 2>&1 | grep round; done
for i in 1 2; do echo "== process $i (one arena)"; REALLOC_ARENA=1 timeout 300 python tools/scan_realloc.py 125 30 6 2>&1 | grep round; done

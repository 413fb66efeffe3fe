python -c "import torch" 2>/dev/null
export SNPGPU_TUNE_LIB=$PWD/tools/ab/libsnpgpu_lo4.so
R=$PWD
cd /tmp; export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  rm -rf /tmp/pm$i
  SWEEP_REPS=2 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm$i -- python $R/tools/scan_sweep.py 125 30 "" > /tmp/o$i.txt 2>&1
  grep "GB/s" /tmp/o$i.txt || tail -5 /tmp/o$i.txt
  python $R/tools/pmc_summary.py /tmp/pm$i 2>&1 | grep -A7 "k_scan_wave<false, 0>" | head -9
done

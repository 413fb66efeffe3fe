#!/bin/sh
# Round profile on the GPU box: the bench line, the rocprofv3 kernel trace of the same command, and the PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ counters, separate runs) -> gpurun_out/<round>/ ; copy what is to be judged into profiles/<round>/.
# Usage: sh tools/profile_round.sh r6
round=${1:-r6}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$round
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
short="--no-live-traffic --cpu-samples 0 --skip-aux --e2e-files 0 --site-files 0 --pipeline-files 0 --shape-samples 0 --skip-call-variants"      # the timed step only (20 steps + 3 warm-up launches): every k_scan_wave<false,0> launch in this trace is a headline launch, so its average is the roofline's
rows="--no-live-traffic --steps 2 --warmup 1 --cpu-samples 0 --skip-aux --skip-separate-steps"                         # the side rows (pipeline from files, end to end, site calling, scan shapes): which kernels they spend their device time in
pmc="--no-live-traffic --steps 3 --warmup 1 --cpu-samples 0 --skip-secondary --skip-aux --e2e-files 0 --site-files 0 --pipeline-files 0 --shape-samples 0 --skip-call-variants"
python $root/bench.py --detail "$out/bench_n1_detail.json" > "$out/bench_n1.json" 2> "$out/bench_n1.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python $root/bench.py $short --detail "$out/trace_bench_detail.json" > "$out/trace_bench.json" 2> "$out/trace.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_rows" -- python $root/bench.py $rows --detail "" > "$out/trace_rows_bench.json" 2> "$out/trace_rows.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python $root/bench.py $pmc --detail "" > /dev/null 2> "$out/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python $root/bench.py $pmc --detail "" > /dev/null 2> "$out/pmc_write.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$out/pmc_sq" -- python $root/bench.py $pmc --detail "" > /dev/null 2> "$out/pmc_sq.err"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$out/pmc_sq2" -- python $root/bench.py $pmc --detail "" > /dev/null 2> "$out/pmc_sq2.err"
# K2 with per-site count records (the call_variants rows): FETCH_SIZE / WRITE_SIZE of k_call_lanes<..., true>
pmc_cv="--no-live-traffic --steps 1 --warmup 1 --cpu-samples 0 --skip-secondary --skip-aux --e2e-files 0 --site-files 0 --pipeline-files 0 --shape-samples 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch_cv" -- python $root/bench.py $pmc_cv --detail "" > /dev/null 2> "$out/pmc_fetch_cv.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write_cv" -- python $root/bench.py $pmc_cv --detail "" > /dev/null 2> "$out/pmc_write_cv.err"
# the one job (hot_path_batch) on 32 samples, half of them resident (the scattered-rows path too): which kernels it launches — no at::native::index* among them
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_job" -- python $root/tools/pipeline_time.py --samples 32 --resident-frac 0.5 > "$out/hot_path_batch_trace.json" 2> "$out/trace_job.err"
find "$out/trace_job" -name "*kernel_stats.csv" -exec cp {} "$out/rocprofv3_kernel_stats_hot_path_batch.csv" \;
( echo "kernels of the hot_path_batch trace whose name contains 'index' (ATen's advanced-indexing kernels are at::native::index_*):"; grep -i "index" "$out/rocprofv3_kernel_stats_hot_path_batch.csv" | cut -d, -f1 | sed 's/^/  /'; echo "(k_lines_index is the library's own line index; no at::native::index* line above = none was launched)"; echo "ATen kernels of any kind in the trace:"; grep -c "at::native" "$out/rocprofv3_kernel_stats_hot_path_batch.csv" ) > "$out/hot_path_batch_aten_kernels.txt" 2>&1
rm -rf "$out/trace_job"
# FETCH_SIZE / WRITE_SIZE on known byte counts
sh $root/tools/fetch_calib.sh 8 > "$out/fetch_calibration_raw.txt" 2>&1
cd "$root"
python tools/pmc_summary.py "$out/pmc_fetch_cv" | grep -A1 "k_call" > "$out/pmc_fetch_size_call_variants_summary.txt"
python tools/pmc_summary.py "$out/pmc_write_cv" | grep -A1 "k_call" > "$out/pmc_write_size_call_variants_summary.txt"
rm -rf "$out/pmc_fetch_cv" "$out/pmc_write_cv"
# phase-1 site calling on a resident sample (the kernels of the pipeline row's ingest): trace averages at three depths + SQ counters at 30x
VS_PMC=30 bash tools/varscan_profile.sh 30 100 8 > "$out/varscan_kernels.txt" 2>&1
python tools/pmc_summary.py "$out/pmc_fetch" > "$out/pmc_fetch_size_summary.txt"
python tools/pmc_summary.py "$out/pmc_write" > "$out/pmc_write_size_summary.txt"
python tools/pmc_summary.py "$out/pmc_sq" > "$out/pmc_sq_summary.txt"
python tools/pmc_summary.py "$out/pmc_sq2" > "$out/pmc_sq2_summary.txt"
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/rocprofv3_kernel_stats_bench.csv" \;
find "$out/trace_rows" -name "*kernel_stats.csv" -exec cp {} "$out/rocprofv3_kernel_stats_side_rows.csv" \;
# keep the merge small: the raw per-dispatch CSVs stay on the box
rm -rf "$out/trace" "$out/trace_rows" "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_sq" "$out/pmc_sq2"
ls -la "$out"
head -12 "$out/rocprofv3_kernel_stats_bench.csv"
grep -A3 "k_scan_wave<false, 0>" "$out/pmc_fetch_size_summary.txt" "$out/pmc_write_size_summary.txt" | head -20
grep -A9 "k_scan_wave<false, 0>" "$out/pmc_sq_summary.txt" "$out/pmc_sq2_summary.txt" | head -40
tail -c 1500 "$out/bench_n1.json"

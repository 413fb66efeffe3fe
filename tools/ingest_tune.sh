#!/bin/bash
# Reader-thread / staging-buffer sweep of the resident ingest (development build with -DSNPGPU_TUNING; rebuilds the product
# library afterwards).  Usage on the GPU box: tools/ingest_tune.sh > gpurun_out/ingest_tune.log
set -e
cd "$(dirname "$0")/.."
SNPGPU_TUNING=1 python -m snp_pipeline_amd.build --force >/dev/null 2>&1
for cfg in "8 4" "12 4" "16 4" "16 8" "24 8" "8 12"; do
    set -- $cfg
    echo "readers=$1 extra_staging=$2"
    SNPGPU_INGEST_READERS=$1 SNPGPU_INGEST_EXTRA_STAGING=$2 python tools/pipeline_time.py --samples ${SAMPLES:-64} --runs 2 --no-vcf 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['runs']: print('   total %.3f  ingest %.3f (%.1f GB/s) alloc %.3f wait_readers %.3f  reading %.2f waiting %.2f' % (r['seconds'], r['ingest']['seconds'], r['file_bytes']/r['ingest']['seconds']/1e9, r['ingest']['allocating'], r['ingest']['waiting_for_readers'], r['ingest']['reader_seconds_reading'], r['ingest']['reader_seconds_waiting']))
print('   pinned h2d %.1f GB/s' % d['pinned_h2d_gb_per_sec'])
"
done
python -m snp_pipeline_amd.build --force >/dev/null 2>&1

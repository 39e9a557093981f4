#!/bin/bash
# Re-collect the rocprofv3 evidence of profiles/ on the GPU box (run through gpurun from the repo root):
#   bash tools/refresh_profiles.sh
# Writes under gpurun_out/prof/; tools/prof_summary.py / tools/pmc_traffic.py turn the CSVs into profiles/r01_*.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
rm -rf "$OUT" && mkdir -p "$OUT"
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-train"
python $R/bench.py --steps 100 --warmup 10 > $OUT/bench_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $R/bench.py $ARGS > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python $R/bench.py $ARGS > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head -20
cat $OUT/bench_line.json

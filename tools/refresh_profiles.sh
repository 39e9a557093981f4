#!/bin/bash
# Re-collect the rocprofv3 evidence of profiles/ on the GPU box (run through gpurun from the repo root):
#   ROUND=r03 bash tools/refresh_profiles.sh     (after tools/build_diag_libs.sh, for the wait statistics)
# Writes under gpurun_out/prof_$ROUND/; tools/make_profiles.py turns that directory into profiles/$ROUND_*.
# Counters are collected in their own passes (--kernel-trace + --pmc only), as the pool requires.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
ROUND=${ROUND:-r04}
# the diagnostic builds must carry the ABI of the shipped library (built together by tools/build_diag_libs.sh in the
# build container, BEFORE the gpurun call): a stale one would leave tracebacks where measurements belong
python $R/tools/check_diag_abi.py || { echo "diag_libs/ do not match liblss_hip.so: run tools/build_diag_libs.sh READSONLY ONEREAD KS_NOREAD KS_NOMFMA KS_WPRE_ALL first"; exit 1; }
OUT=$R/gpurun_out/prof_$ROUND
rm -rf "$OUT" && mkdir -p "$OUT"
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-train --no-two-streams"  # profiled runs: the single-stream loop only
cd $R
python bench.py --steps 100 --warmup 10 > $OUT/bench_line.json 2> $OUT/bench_line.err
echo "bench line done" 
python bench.py --workload hires --steps 50 --warmup 10 --no-train --no-cpu-baseline > $OUT/bench_hires.json 2>/dev/null
python bench.py --precision fp32 --steps 20 --warmup 5 --no-train --no-cpu-baseline > $OUT/bench_fp32.json 2>/dev/null
python bench.py --workload config1 --steps 100 --warmup 10 --no-train > $OUT/bench_config1.json 2>/dev/null
LSS_BENCH_REHEARSE=1 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_rehearse2.txt 2>/dev/null
python tools/bench_kernels.py --only conv > $OUT/bk_conv.txt 2>&1
python tools/bench_kernels.py --only stamps > $OUT/bk_stamps.txt 2>&1
python tools/bench_kernels.py --only l1 > $OUT/bk_l1.txt 2>&1
python tools/bench_l1.py --stamps > $OUT/l1_stamps.txt 2>&1
python tools/bench_ring.py > $OUT/ring_microbench.txt 2>&1
[ -f diag_libs/liblss_STATS.so ] && LSS_HIP_LIB=$R/diag_libs/liblss_STATS.so python tools/ring_stats.py > $OUT/ring_wait_stats.txt 2>&1
[ -f diag_libs/liblss_NOBLEND.so ] && LSS_HIP_LIB=$R/diag_libs/liblss_NOBLEND.so python tools/bench_ring.py > $OUT/ring_microbench_noblend.txt 2>&1
[ -f diag_libs/liblss_NOWDMA.so ] && LSS_HIP_LIB=$R/diag_libs/liblss_NOWDMA.so python tools/bench_ring.py > $OUT/ring_microbench_nowdma.txt 2>&1
[ -f diag_libs/liblss_READSONLY.so ] && LSS_HIP_LIB=$R/diag_libs/liblss_READSONLY.so python tools/bench_ring.py > $OUT/ring_microbench_readsonly.txt 2>&1
[ -f diag_libs/liblss_ONEREAD.so ] && LSS_HIP_LIB=$R/diag_libs/liblss_ONEREAD.so python tools/bench_ring.py > $OUT/ring_microbench_oneread.txt 2>&1
python tools/bench_wgrad.py > $OUT/wgrad_microbench.txt 2>&1
python tools/bench_ks.py > $OUT/ks_microbench.txt 2>&1
python tools/bench_ks.py --stamps > $OUT/ks_stamps.txt 2>&1
for v in KS_NOREAD KS_NOMFMA KS_WPRE_ALL; do
  [ -f diag_libs/liblss_$v.so ] && { echo "--- $v"; LSS_HIP_LIB=$R/diag_libs/liblss_$v.so python tools/bench_ks.py --stamps; LSS_HIP_LIB=$R/diag_libs/liblss_$v.so python tools/bench_ks.py; } >> $OUT/ks_stamps_diag.txt 2>&1
done
python tools/graph_node_cost.py > $OUT/graph_node_cost.txt 2>&1
echo "microbenches done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $R/bench.py $ARGS > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python $R/bench.py $ARGS > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/sq -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-two-streams > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq_ring -- python $R/tools/bench_ring.py --rounds 2 --iters 3 > $OUT/sq_ring.log 2>&1
echo "pmc passes done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_hires -- python $R/bench.py --workload hires $ARGS > $OUT/kt_hires.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_fp32 -- python $R/bench.py --precision fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-two-streams > $OUT/kt_fp32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_train -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-two-streams --train-steps 10 > $OUT/kt_train.log 2>&1
cd $R
python tools/train_step_trace.py $(find $OUT/kt_train -name "*kernel_trace.csv" | head -1) 70 > $OUT/train_step_trace.txt 2>&1
find $OUT -name "*.csv" | wc -l
tail -c 600 $OUT/bench_line.json

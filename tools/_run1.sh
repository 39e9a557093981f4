set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py tests/test_bench_config_gpu.py tests/test_conv_ring_gpu.py -m gpu -x -q > gpurun_out/r3_t7.log 2>&1; tail -5 gpurun_out/r3_t7.log
python bench.py --steps 20 --warmup 5 --no-train > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; python -c "
import json; d=json.load(open('gpurun_out/r3_bench1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['dominant_kernel']['avg_us'], d['roofline_l1']['level_us'], d['levels']['two_batches_in_flight_fps'])"
LSS_CONV_RING=0 python bench.py --steps 20 --warmup 5 --no-train --no-cpu-baseline > gpurun_out/r3_bench1_noring.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3_bench1_noring.json')); print('noring', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ring1 -- python $GRAFT_REPO_ROOT/tools/bench_ring.py --rounds 2 --iters 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_ring1.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_sq_summary.py $(find gpurun_out/prof_ring1 -name "*counter_collection.csv") $(find gpurun_out/prof_ring1 -name "*kernel_trace.csv") > gpurun_out/prof_ring1_summary.txt 2>&1; cat gpurun_out/prof_ring1_summary.txt | cut -c1-250

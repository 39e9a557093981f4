set -o pipefail
cd $GRAFT_REPO_ROOT
python bench.py --steps 50 --warmup 10 --no-train --no-cpu-baseline > gpurun_out/r3_bench2.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3_bench2.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['dominant_kernel']['avg_us'], d['roofline_l1']['level_us'], d['levels']['two_batches_in_flight_fps'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03a/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-two-streams > $GRAFT_REPO_ROOT/gpurun_out/prof_r03a/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find gpurun_out/prof_r03a/kt -name "*kernel_trace.csv") > gpurun_out/prof_r03a_summary.txt 2>&1; head -40 gpurun_out/prof_r03a_summary.txt | cut -c1-200

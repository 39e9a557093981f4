set -o pipefail
cd /tmp && export TMPDIR=/tmp
for b in 1 8; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03b/kt$b -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-two-streams > $GRAFT_REPO_ROOT/gpurun_out/prof_r03b/kt$b.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find $GRAFT_REPO_ROOT/gpurun_out/prof_r03b/kt$b -name "*kernel_trace.csv") > $GRAFT_REPO_ROOT/gpurun_out/prof_r03b_b$b.txt 2>&1; echo "batch $b"; head -14 $GRAFT_REPO_ROOT/gpurun_out/prof_r03b_b$b.txt | cut -c1-130
done

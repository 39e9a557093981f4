#!/bin/bash
# Same-box A/B of one environment switch through bench.py (alternating runs):  bash tools/ab_env.sh LSS_SPLAT_DIRECT [rounds]
VAR=${1:-LSS_SPLAT_DIRECT}; ROUNDS=${2:-3}
for i in $(seq $ROUNDS); do
for v in 0 1; do
env $VAR=$v python bench.py --steps 100 --warmup 10 --no-train --no-cpu-baseline --no-two-streams 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', round(d['value']), 'fps  %.4f ms/step  level %.1f us  conv group %.1f us  dominant %.1f us' % (d['ms_per_step'], d['roofline_l1']['level_us'], d['roofline']['avg_us']*16, d['roofline']['dominant_kernel']['avg_us']))"
done; done

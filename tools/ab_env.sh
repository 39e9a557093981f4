for i in 1 2 3; do
for v in 0 1; do
LSS_K2_XCD=$v python bench.py --steps 100 --warmup 10 --no-train --no-cpu-baseline --no-two-streams 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xcd=$v', round(d['value']), 'fps  level %.1f us  conv %.1f' % (d['roofline_l1']['level_us'], d['roofline']['avg_us']*16))"
done; done

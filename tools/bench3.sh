#!/bin/bash
# three bench lines (C2, one rank's share of C2 at 8 GPUs, one rank's share of C4) -> launch ms / frac / frames/s
for args in "" "--samples 25" "--workload c4 --samples 125"; do
  out=$(python bench.py --no-alt --no-cpu-baseline --steps 6 --warmup 2 $args 2>/dev/null | tail -1)
  echo "[$args] $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('launch_ms %.2f frac %.4f frames/s %.0f ms/step %.1f' % (r['avg_launch_us']/1e3, r['frac'], d['value'], d['ms_per_step']))")"
done

#!/bin/bash
# A/B timing of library variants on the GPU box: tools/ab_bench.sh <tag>=<lib.so>[:ENV=VAL...] ... [-- bench args]
# prints avg launch us / frames/s per variant (primary precision only, no CPU leg).
args=()
variants=()
seen_dd=0
for a in "$@"; do
    if [ "$a" == "--" ]; then seen_dd=1; continue; fi
    if [ $seen_dd == 1 ]; then args+=("$a"); else variants+=("$a"); fi
done
for v in "${variants[@]}"; do
    tag=${v%%=*}; rest=${v#*=}
    lib=${rest%%:*}
    envs=""
    if [[ "$rest" == *:* ]]; then envs=$(echo "${rest#*:}" | tr ':' ' '); fi
    out=$(env $envs VF_LIBRARY=$lib python bench.py --no-alt --no-cpu-baseline --steps 6 --warmup 2 "${args[@]}" 2>/dev/null | tail -1)
    echo "$tag $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('launch_us %.0f frac %.4f frames/s %.0f ms/step %.1f' % (r['avg_launch_us'], r['frac'], d['value'], d['ms_per_step']))")"
done

#!/bin/bash
# Alternating bench.py runs of one config under several hiprtc option sets (each may contain spaces), on one box.
# usage: tools/sched_ab.sh <config> <rounds> "<flags>" ...      ("-" = default)
cfg=$1; rounds=$2; shift 2
for r in $(seq $rounds); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then fl=""; else fl="$v"; fi
    line=$(GAME_HIP_FLAGS="$fl" timeout -k 10 120 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu --no-others 2>/dev/null | tail -1)
    echo "$cfg [$v] $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], r.get("avg_launch_ms", r.get("avg_step_ms")))')"
  done
done

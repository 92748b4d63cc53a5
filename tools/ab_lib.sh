#!/bin/bash
# Alternating bench.py runs of two builds of the library on one box: what a change in game_device.h is worth.
# Build the two, copy them to game-engine_amd/libgame_hip_old.so and game-engine_amd/libgame_hip_new.so (they travel with
# the tree; delete them and rebuild afterwards), then on the GPU box:
# usage: tools/ab_lib.sh <rounds> <config>...
rounds=$1; shift
for r in $(seq $rounds); do
  for v in old new; do
    cp game-engine_amd/libgame_hip_$v.so game-engine_amd/libgame_hip.so
    for cfg in "$@"; do
      line=$(timeout -k 10 120 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu --no-others 2>/dev/null | tail -1)
      echo "$cfg [$v] $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], r.get("avg_launch_ms", r.get("avg_step_ms")))')"
    done
  done
done

#!/bin/bash
# Pipelined sequence under other tile orders / pattern-cell sizes: in a sequence the tail of a launch is filled by
# the next scene's kernels, so the locality given up for the heaviest-first order may be worth more than the tail.
set -e
run() {
  echo -n "[$*] "
  env "$@" timeout -k 10 200 python3 bench.py --no-variants --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['conv_ms_per_scene'])"
}
run M3D_TILE_ORDER=1
run M3D_TILE_ORDER=0
run M3D_TILE_ORDER=0 M3D_SCHED_MODES=1539,3843,3843,0
run M3D_TILE_ORDER=0 M3D_SCHED_MODES=1795,3843,3843,0
run M3D_TILE_ORDER=1 M3D_SCHED_MODES=1795,3843,3843,0
run M3D_TILE_ORDER=0 M3D_SCHED_MODES=1539,1539,3843,0
run M3D_TILE_ORDER=1

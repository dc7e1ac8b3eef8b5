#!/bin/bash
# Pipelined throughput (tools/pipe_times.py) for a list of "M3D_SCHED_MODES:M3D_TILE_ORDER" settings.
# usage: bash tools/pipe_sweep.sh <precision> <modes:order> ...
set -e -o pipefail
prec=$1; shift
for s in "$@"; do
  modes=${s%%:*}; order=${s##*:}
  echo "== modes=$modes tile_order=$order"
  M3D_SCHED_MODES=$modes M3D_TILE_ORDER=$order timeout -k 10 120 python3 tools/pipe_times.py $prec 30 2>/dev/null | grep "one at"
done

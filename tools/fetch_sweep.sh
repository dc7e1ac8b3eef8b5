#!/bin/bash
# Duration and FETCH_SIZE of the level-0 96 -> 96 submanifold convolution for a list of
# "M3D_SCHED_MODES:M3D_TILE_ORDER" settings.  usage: bash tools/fetch_sweep.sh <precision> <modes:order> ...
set -e -o pipefail
prec=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for s in "$@"; do
  modes=${s%%:*}; order=${s##*:}
  export M3D_SCHED_MODES=$modes M3D_TILE_ORDER=$order PREC=$prec
  echo "== modes=$modes tile_order=$order"
  timeout -k 10 120 python3 $root/tools/level_conv.py ${LCONV:-0 96 96} 2>/dev/null | grep level
  rm -rf /tmp/fs
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fs -o fs -- python3 $root/tools/level_conv.py ${LCONV:-0 96 96} > /dev/null 2>&1
  python3 - <<'PY'
import csv
v = [float(r["Counter_Value"]) for r in csv.DictReader(open("/tmp/fs/fs_counter_collection.csv")) if "conv_igemm_dma" in r["Kernel_Name"]]
print(f"   FETCH_SIZE {sum(v) / len(v) / 1024:.1f} MiB per launch (x2 for bytes), {len(v)} launches")
PY
done

#!/bin/bash
# Forced tile plans (wm,wn,nt,nz) against the library's own on the small levels.  usage: bash tools/plan_sweep.sh <precision>
set -e -o pipefail
export PREC=$1
for spec in "2 128 128" "2 256 128" "3 256 256" "3 512 256" "4 256 256" "1 64 64" "2 64 128"; do
  timeout -k 10 200 python3 tools/level_conv.py $spec - 1,4,1,1 2,2,2,1 2,2,1,1 1,2,1,1 4,1,4,1 2,2,2,2 1,4,1,2 1,4,1,3 2>/dev/null | grep level
done

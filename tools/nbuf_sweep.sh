#!/bin/bash
# Ring depth (M3D_NBUF) on the small levels, where a launch is bound by the latency of its stages, not by throughput
set -e
for nb in 2 3 4; do
  for c in "4 256 256" "3 128 128" "3 256 256" "2 64 64" "2 128 128" "1 32 32"; do
    echo -n "NBUF=$nb "
    PREC=bf16x6 M3D_NBUF=$nb timeout -k 10 120 python3 tools/level_conv.py $c 2>/dev/null | grep level
  done
done

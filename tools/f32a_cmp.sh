#!/bin/bash
# bf16x6 convolutions of the benchmark scene: pre-split (sf32) input against plain f32 rows split after the fragment read
set -e -o pipefail
for spec in "0 96 96" "0 128 96" "0 32 32" "1 96 96" "1 32 32" "1 192 96" "2 64 64" "2 128 128" "3 128 128" "0d 32 32" "0u 96 96" "0h 96 768"; do
  PREC=bf16x6 F32IN=0 timeout -k 10 120 python3 tools/level_conv.py $spec 2>/dev/null | grep level | sed 's/^/sf32 /'
  PREC=bf16x6 F32IN=1 timeout -k 10 120 python3 tools/level_conv.py $spec 2>/dev/null | grep level | sed 's/^/f32  /'
done

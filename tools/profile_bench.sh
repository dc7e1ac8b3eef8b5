#!/bin/bash
# Kernel-trace statistics and the two HBM-traffic counter passes of one bench.py configuration, each in a run of its
# own, with the scenes processed one at a time (--no-pipeline): kernel durations are then those of a kernel that has
# the GPU to itself, which is what bench.py's roofline object measures with HIP events.  usage (on the GPU box, from the repository root): bash tools/profile_bench.sh <tag> [bench.py arguments]
# Leaves gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_fetch.csv,pmc_write.csv}.
set -e -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/kt -o kt -- \
  python3 "$root/bench.py" --no-cpu-baseline --no-variants --no-pipeline --steps 20 --warmup 5 "$@" > "$out/bench_line.json" 2> "$out/kt.err"
cp /tmp/prof_$tag/kt/kt_kernel_stats.csv "$out/kernel_stats.csv"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_$tag/pf -o pf -- \
  python3 "$root/bench.py" --no-cpu-baseline --no-variants --no-pipeline --steps 3 --warmup 1 "$@" > /dev/null 2> "$out/pf.err"
cp /tmp/prof_$tag/pf/pf_counter_collection.csv "$out/pmc_fetch.csv"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_$tag/pw -o pw -- \
  python3 "$root/bench.py" --no-cpu-baseline --no-variants --no-pipeline --steps 3 --warmup 1 "$@" > /dev/null 2> "$out/pw.err"
cp /tmp/prof_$tag/pw/pw_counter_collection.csv "$out/pmc_write.csv"
ls -la "$out"

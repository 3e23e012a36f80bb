#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace of the bench command;
# the stats CSVs land in gpurun_out/prof_<tag>/ and are copied by hand into
# profiles/ afterwards.  Usage: tools/gpu_profile.sh <tag> [bench args...]
set -eo pipefail
tag=${1:-r1}
shift || true
out=$(pwd)/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o trace -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench_stdout.log" 2> "$out/rocprof_stderr.log" || { tail -20 "$out/rocprof_stderr.log"; exit 1; }
tail -2 "$out/bench_stdout.log"
f=$(find "$out" -name '*kernel_stats.csv' | sort | sed -n 1p)
if [ -n "$f" ]; then sed -n 1,8p "$f"; else echo "no kernel_stats.csv under $out"; ls -R "$out"; fi

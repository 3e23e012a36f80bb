#!/bin/bash
# Run on the GPU box: perf_ab lines for the library in the tree and for every build given (name=path ...).
# Usage: WORKLOADS="c3 boxes" tools/gpu_ab_many.sh name=path [name=path ...]
cd "$(dirname "$0")/.."
tools/perf_ab.sh tree
for kv in "$@"; do
  RACER_TRACER_AMD_LIB=$PWD/${kv#*=} tools/perf_ab.sh "${kv%%=*}"
done

#!/bin/bash
# Run on the GPU box: bench lines for c3/c2/c4.
# Usage: tools/perf_ab.sh "<tag>" [ENV=VAL ...]
# ENV=VAL knobs (RT_DBG0..3, RT_POOL_CHUNK, RT_POOL_BLOCKS_PER_CU, RT_BVH_LDS) only exist in the developer
# build: when any is given, the run uses build/libracer_tracer_amd_dev.so (make -C racer-tracer_amd dev-lib).
# RACER_TRACER_AMD_LIB=... selects any other build of the library.
tag=$1; shift
for kv in "$@"; do
  export "$kv"
  case "$kv" in RT_*) [ -z "$RACER_TRACER_AMD_LIB" ] && export RACER_TRACER_AMD_LIB=$(dirname "$0")/../racer-tracer_amd/build/libracer_tracer_amd_dev.so;; esac
done
for w in ${WORKLOADS:-c3 c2 c4}; do
  timeout -k 10 90 python3 bench.py --workload $w --no-cpu-baseline --pmc none --no-host-delivery --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$tag $w %.0f Msamples/s  %.2f ms  %.2f Gseg/s' % (d['value'], d['ms_per_step'], d['roofline']['gsegments_per_s']))"
done

#!/bin/bash
# Run on the GPU box: where do the waves of the trace kernel spend their time?
# Uses the -DRT_PROFILE_REGIONS build (make -C racer-tracer_amd profile-lib, built
# in the container so it travels with the snapshot... build/ is not ignored by gpurun).
# Prints, per workload, the share of wave-resident shader cycles per region of the
# path loop.  Wall-clock of the wave, so latency-heavy regions weigh more than their
# share of issued instructions.
set -eo pipefail
cd "$(dirname "$0")/.."
lib=$PWD/racer-tracer_amd/build/libracer_tracer_amd_regions.so
[ -f "$lib" ] || { echo "missing $lib (make -C racer-tracer_amd profile-lib)"; exit 1; }
for w in ${@:-c3 c2 c4}; do
  echo "== $w"
  RACER_TRACER_AMD_LIB=$lib timeout -k 10 120 python3 bench.py --workload $w --no-cpu-baseline --pmc none --no-host-delivery --steps 1 --warmup 1 2>&1 | grep -E "^region" || true
done

#!/bin/bash
# Launch-bound (waves per SIMD) sweep per variant family: build/libracer_tracer_amd_ab_<name>.so from
#   make -C racer-tracer_amd ab-lib AB_NAME=<name> AB_FLAGS=-DRT_OCC_<FAMILY>=<n>
cd "$(dirname "$0")/.."
run() { # tag lib workloads...
  tag=$1; lib=$2; shift 2
  for w in "$@"; do
    RACER_TRACER_AMD_LIB=$lib timeout -k 10 100 python3 bench.py --workload $w --no-cpu-baseline --pmc none --no-host-delivery --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-10s %-8s %.2f ms  %.2f Gseg/s' % ('$tag', '$w', d['ms_per_step'], d['roofline']['gsegments_per_s']))"
  done
}
B=$PWD/racer-tracer_amd/build
run shipped $PWD/racer-tracer_amd/lib/libracer_tracer_amd.so c2 boxes c4 emissive random
run spec6 $B/libracer_tracer_amd_ab_spec6.so c2
run spec4 $B/libracer_tracer_amd_ab_spec4.so c2
run any5 $B/libracer_tracer_amd_ab_any5.so boxes
run any3 $B/libracer_tracer_amd_ab_any3.so boxes
run tex3 $B/libracer_tracer_amd_ab_tex3.so c4
run tex5 $B/libracer_tracer_amd_ab_tex5.so c4
run texany3 $B/libracer_tracer_amd_ab_texany3.so emissive
run texany5 $B/libracer_tracer_amd_ab_texany5.so emissive
run texbvh3 $B/libracer_tracer_amd_ab_texbvh3.so random
run texbvh5 $B/libracer_tracer_amd_ab_texbvh5.so random

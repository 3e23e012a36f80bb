#!/bin/bash
# Run on the GPU box: GPU tests, then bench lines + PMC passes + region profile of the current tree.
# Usage: [NOPMC=1] [NOREGIONS=1] [WORKLOADS="c4"] tools/gpu_round_baseline.sh <tag> [notests]
tag=${1:-r2}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
if [ "$2" != "notests" ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/${tag}_tests.log 2>&1
  rc=$?
  tail -15 gpurun_out/${tag}_tests.log
  [ $rc -ge 124 ] && exit $rc
fi
for w in ${WORKLOADS:-c3 c2 c4}; do
  timeout -k 10 200 python3 bench.py --workload $w --steps 5 --warmup 2 $([ $w != c3 ] && echo --no-cpu-baseline) > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err || exit $?
  python3 -c "
import json
d=json.load(open('gpurun_out/${tag}_bench_$w.json'))
print('$tag $w %.0f Msamples/s  %.2f ms  %.2f Gseg/s' % (d['value'], d['ms_per_step'], d['roofline']['gsegments_per_s']))"
done
[ -n "$NOPMC" ] && { [ -z "$NOREGIONS" ] && timeout -k 10 300 tools/region_profile.sh ${WORKLOADS:-c3 c2 c4} 2>&1 | tee gpurun_out/${tag}_regions.log; exit 0; }
for w in c3 c2 c4; do
  timeout -k 10 400 tools/gpu_pmc.sh ${tag}_$w standard -- --workload $w > gpurun_out/${tag}_pmc_$w.log 2>&1 || { tail -5 gpurun_out/${tag}_pmc_$w.log; exit 1; }
  echo "pmc $w done"
done
timeout -k 10 300 tools/region_profile.sh c3 c2 c4 > gpurun_out/${tag}_regions.log 2>&1 || exit $?
cat gpurun_out/${tag}_regions.log

#!/bin/bash
# Run on the GPU box: bench lines for c3/c2/c4 under the given env settings.
# Usage: tools/perf_ab.sh "<tag>" [ENV=VAL ...]
tag=$1; shift
for kv in "$@"; do export "$kv"; done
for w in c3 c2 c4; do
  timeout -k 10 90 python3 bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$tag $w %.0f Msamples/s  %.2f ms  %.2f Gseg/s' % (d['value'], d['ms_per_step'], d['roofline']['gsegments_per_s']))"
done

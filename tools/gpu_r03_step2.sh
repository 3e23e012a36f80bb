#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03_step2
timeout -k 10 300 python3 -m pytest tests/test_gpu_delivery.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r03_step2/tests.txt
echo "== bench default"; (time timeout -k 10 600 python3 bench.py > gpurun_out/r03_step2/bench_c3.json 2> gpurun_out/r03_step2/bench_c3.err) 2>&1 | tail -4; tail -12 gpurun_out/r03_step2/bench_c3.err; python3 -c "
import json
d=json.load(open('gpurun_out/r03_step2/bench_c3.json'))
print(json.dumps({k:d[k] for k in ('value','ms_per_step','host_delivered')}, indent=1))
print(json.dumps(d['roofline'], indent=1)[:3000])
print(d.get('cpu_baseline'))"

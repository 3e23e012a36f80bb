#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03_step1
timeout -k 10 600 python3 -m pytest tests/test_gpu_delivery.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/r03_step1/tests.txt
echo "== perf"; tools/perf_ab.sh new | tee gpurun_out/r03_step1/perf.txt
timeout -k 10 120 python3 tools/time_tiles.py 2>&1 | tail -12 | tee gpurun_out/r03_step1/time_tiles.txt

#!/bin/bash
# Run on the GPU box after tools/gpu_finalize_profiles.sh <round> (whose PMC summaries must already be under profiles/):
#   tools/gpu_finalize_more.sh r04
set -o pipefail
round=${1:-r04}
cd "$(dirname "$0")/.."
out=gpurun_out/$round
mkdir -p "$out"
export TMPDIR=/tmp
tools/gpu_extra_measurements.sh $round
# executed-opcode histograms of the instrumented builds (tools/bb_build.sh <variants> in the container first)
SKIP_PMC=1 tools/gpu_opcode_hist.sh $round > "$out/opcode_hist.log" 2>&1
grep -h "^valu\|^salu\|^==" "$out/opcode_hist.log"
timeout -k 10 200 python3 tools/time_scene_create.py "$out/${round}_scene_create.txt" 2>&1 | grep -v amdgpu.ids | grep "create + first"
export BENCH_REHEARSE_ON_ONE_GPU=1
timeout -k 10 300 python3 bench.py --gpus 4 --steps 2 --warmup 1 --spp 64 --no-cpu-baseline > "$out/${round}_bench_rehearsal_gpus4.json" 2> "$out/rehearsal4.err"
timeout -k 10 400 python3 bench.py --gpus 5 --workload c5 --steps 1 --warmup 0 --no-cpu-baseline > "$out/${round}_c5_rehearsal_5ranks.json" 2> "$out/rehearsal5.err"
python3 -c "
import json
for f in ('$out/${round}_bench_rehearsal_gpus4.json', '$out/${round}_c5_rehearsal_5ranks.json'):
    d = json.load(open(f))
    print(d['n_gpus'], 'ranks on one card,', d['config']['workload'], '| gathered frame matches the single-rank frame:', d['gathered_frame_matches_single_rank'])"

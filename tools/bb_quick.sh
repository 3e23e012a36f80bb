#!/bin/bash
# Developer loop (run on the GPU box after tools/bb_build.sh <variants> in the container): instructions per segment and the
# copy / lane-op opcodes of the given workload:variant pairs.
#   tools/bb_quick.sh c3:Li0ELb0ELb0ELb0E random:Li2ELb1ELb1ELb1E
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/bbq
for pair in "$@"; do
  w=${pair%%:*}; v=${pair#*:}
  timeout -k 10 200 python3 tools/bb_profile.py $w $v gpurun_out/bbq/${w}_hist.txt > gpurun_out/bbq/bb_$w.log 2>&1 || { echo "bb $w failed"; tail -5 gpurun_out/bbq/bb_$w.log; continue; }
  echo "== $w"; grep -E "^(valu|salu|lds|vmem) " gpurun_out/bbq/${w}_hist.txt; grep -E "^  (v_readlane_b32|v_writelane_b32|v_mov_b32|v_mov_b64|scratch_\w+) " gpurun_out/bbq/${w}_hist.txt
done

#!/bin/bash
# Run on the GPU box: PMC summaries of the product build and executed-opcode histograms of the instrumented builds
# (tools/bb_build.sh ... in the container first), side by side, for the round given.
#   tools/gpu_opcode_hist.sh r04 [workload:variant ...]
set -o pipefail
cd "$(dirname "$0")/.."
round=${1:-r04}; shift
pairs=${@:-c3:Li0ELb0ELb0ELb0E c4:Li1ELb1ELb1ELb0E random:Li2ELb1ELb1ELb1E boxes:Li2ELb0ELb0ELb0E c2:Li1ELb0ELb1ELb0E}
out=gpurun_out/$round
mkdir -p $out profiles
for pair in $pairs; do
  w=${pair%%:*}; v=${pair#*:}
  echo "== $w ($v)"
  if [ -z "$SKIP_PMC" ]; then
    timeout -k 10 400 python3 tools/pmc.py $out/${round}_${w}_pmc_summary.json --groups all -- --workload $w > $out/pmc_$w.log 2>&1 || echo "pmc $w: rc $?"
    cp $out/${round}_${w}_pmc_summary.json profiles/ 2>/dev/null   # bb_profile reads the summary of the same tree from profiles/
  fi
  timeout -k 10 300 python3 tools/bb_profile.py $w $v $out/${round}_${w}_opcode_hist.txt > $out/bb_$w.log 2>&1 || { echo "bb_profile $w: rc $?"; tail -5 $out/bb_$w.log; }
  head -32 $out/${round}_${w}_opcode_hist.txt
done

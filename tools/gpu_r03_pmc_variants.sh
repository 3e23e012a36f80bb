#!/bin/bash
# PMC summaries of the kernel variants that are not BASELINE configs: random (BVH), cornell_box_boxes and emissive (PRIMS_ANY)
set -o pipefail
cd "$(dirname "$0")/.."
out=gpurun_out/r03_pmc
mkdir -p $out
for w in random boxes emissive; do
  timeout -k 10 300 python3 tools/pmc.py $out/r03_${w}_pmc_summary.json --groups all -- --workload $w > $out/pmc_$w.log 2>&1; echo "$w rc $?"; grep -c mean= $out/pmc_$w.log
done
tail -60 $out/pmc_random.log

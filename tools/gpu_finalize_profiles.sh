#!/bin/bash
# Run on the GPU box: everything profiles/ holds for a round, taken on the current tree.
#   tools/gpu_finalize_profiles.sh r02
# 1. PMC passes (VALU, waits, FETCH_SIZE, WRITE_SIZE) per workload -> gpurun_out/<round>/<w>_pmc_summary.json (stamped)
# 2. bench lines per workload WITH those summaries installed under profiles/ (so the line carries the roofline)
# 3. rocprofv3 --kernel-trace --stats of the C3 bench command -> kernel_stats.csv
# 4. region / lane profile of the -DRT_PROFILE_REGIONS build
# 5. the `random` scene through the BVH and the linear loop, C5 on one card, per-rank shares
# 6. the N = 2 rehearsal of bench.py's self-launch on one card
# Copy gpurun_out/<round>/* into profiles/ afterwards.
set -eo pipefail
round=${1:-r02}
cd "$(dirname "$0")/.."
out=gpurun_out/$round
mkdir -p "$out"
for w in c3 c2 c4; do
  timeout -k 10 400 tools/gpu_pmc.sh ${round}_$w standard -- --workload $w > "$out/pmc_$w.log" 2>&1 || { tail -5 "$out/pmc_$w.log"; exit 1; }
  cp gpurun_out/pmc_${round}_$w/summary.json "$out/${round}_${w}_pmc_summary.json"
  cp "$out/${round}_${w}_pmc_summary.json" profiles/${round}_${w}_pmc_summary.json
  echo "pmc $w done"
done
for w in c3 c2 c4; do
  timeout -k 10 300 python3 bench.py --workload $w --steps 5 --warmup 2 > "$out/${round}_${w}_bench.json" 2> "$out/bench_$w.err"
  python3 -c "
import json
d=json.load(open('$out/${round}_${w}_bench.json'))
r=d['roofline']
print('$w %.0f Msamples/s %.2f ms %.2f Gseg/s valu frac %s lanes %s useful %s traffic %s GB/s' % (d['value'], d['ms_per_step'], r['gsegments_per_s'], r['frac'], r.get('lanes_per_inst'), r.get('useful_lane_frac'), r['traffic']))"
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_c3" -o trace -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > "$out/${round}_c3_bench_under_rocprof.json" 2> "$out/rocprof_c3.err" || { tail -5 "$out/rocprof_c3.err"; exit 1; }
f=$(find "$out/prof_c3" -name '*kernel_stats.csv' | sort | sed -n 1p)
cp "$f" "$out/${round}_c3_kernel_stats.csv"
sed -n 1,6p "$out/${round}_c3_kernel_stats.csv"
rm -rf "$out/prof_c3"
timeout -k 10 300 tools/region_profile.sh c3 c2 c4 > "$out/${round}_region_cycles.txt" 2>&1
grep -c region "$out/${round}_region_cycles.txt"
timeout -k 10 300 python3 tools/perf_random.py 64 2>&1 | grep -v amdgpu.ids | tee "$out/${round}_random_scene.txt"
timeout -k 10 200 python3 tools/quick_perf_yml.py 128 2>&1 | grep -v amdgpu.ids | tee "$out/${round}_all_scenes.txt"
timeout -k 10 200 python3 tools/c5_check.py 2>&1 | grep -v amdgpu.ids | tee "$out/${round}_c5_check.txt"
timeout -k 10 200 python3 tools/strip_share.py 2>&1 | grep -v amdgpu.ids | tee "$out/${round}_strip_share.txt"
BENCH_REHEARSE_ON_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > "$out/${round}_bench_rehearsal_gpus2.json" 2> "$out/rehearsal.err"
python3 -c "
import json
d=json.load(open('$out/${round}_bench_rehearsal_gpus2.json'))
print('rehearsal n_gpus', d['n_gpus'], 'frame matches single rank:', d['rehearsal_frame_matches_single_rank'])"

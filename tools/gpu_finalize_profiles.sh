#!/bin/bash
# Run on the GPU box: everything profiles/ holds for a round, taken on the current tree.
#   tools/gpu_finalize_profiles.sh r03
# 1. PMC passes (tools/pmc.py, every counter group) per workload, BASELINE configs and the other kernel variants
#    (random = BVH, boxes / emissive = PRIMS_ANY) -> <round>_<w>_pmc_summary.json (stamped with the kernel sources' hash)
# 2. bench lines per BASELINE workload (live counters, host-delivered rates, CPU baseline on c3)
# 3. rocprofv3 --kernel-trace --stats of the C3 bench command -> kernel_stats.csv
# 4. region / lane profile of the -DRT_PROFILE_REGIONS build (if it has been built: make -C racer-tracer_amd profile-lib)
# 5. the `random` scene through the BVH and the linear loop, every shipped scene, C5 on one card, per-rank shares
# 6. bench.py's N > 1 path on one card: the gloo rehearsal with 2 ranks, and RCCL with one rank (--force-dist)
# Second call (the box's time limit): tools/gpu_finalize_more.sh <round> — kernel statistics of the other workloads, the
# sustained run, seeds, executed-opcode histograms, scene-create latency, the 4- and 5-rank rehearsals.
# Copy gpurun_out/<round>/<round>_* into profiles/ afterwards.
set -o pipefail
round=${1:-r03}
cd "$(dirname "$0")/.."
out=gpurun_out/$round
mkdir -p "$out"
export TMPDIR=/tmp
for w in c3 c2 c4 random boxes emissive; do
  timeout -k 10 400 python3 tools/pmc.py "$out/${round}_${w}_pmc_summary.json" --groups all -- --workload $w > "$out/pmc_$w.log" 2>&1 || { tail -5 "$out/pmc_$w.log"; exit 1; }
  rm -rf "$out/${round}_${w}_pmc_summary_passes"
  echo "pmc $w done"
done
for w in c3 c2 c4 boxes emissive random; do
  extra="--no-cpu-baseline"; [ $w == c3 ] && extra=""
  timeout -k 10 400 python3 bench.py --workload $w $extra > "$out/${round}_${w}_bench.json" 2> "$out/bench_$w.err" || { tail -5 "$out/bench_$w.err"; exit 1; }
  python3 -c "
import json
d=json.load(open('$out/${round}_${w}_bench.json'))
r=d['roofline']; h=d.get('host_delivered',{})
print('$w %.0f Msamples/s %.2f ms %.2f Gseg/s | f64 flop frac %s issue frac %s busy %s lanes %s | traffic %s GB/s | rt_render %s ms rt_render_frame %s ms' % (d['value'], d['ms_per_step'], r['gsegments_per_s'], r['frac'], r.get('issue_frac'), r.get('valu_busy'), r.get('lanes_per_inst'), r['traffic'], h.get('rt_render',{}).get('ms'), h.get('rt_render_frame',{}).get('ms')))"
done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$out/prof_c3" -o trace -- python3 "$OLDPWD/bench.py" --no-cpu-baseline --pmc none --no-host-delivery --steps 10 --warmup 2 > "$OLDPWD/$out/${round}_c3_bench_under_rocprof.json" 2> "$OLDPWD/$out/rocprof_c3.err") || { tail -5 "$out/rocprof_c3.err"; exit 1; }
f=$(find "$out/prof_c3" -name '*kernel_stats.csv' | sort | sed -n 1p)
cp "$f" "$out/${round}_c3_kernel_stats.csv"
sed -n 1,6p "$out/${round}_c3_kernel_stats.csv"
rm -rf "$out/prof_c3"
if [ -f racer-tracer_amd/build/libracer_tracer_amd_regions.so ]; then
  timeout -k 10 400 tools/region_profile.sh c3 c2 c4 boxes random > "$out/${round}_region_cycles.txt" 2>&1
  grep -c region "$out/${round}_region_cycles.txt"
fi
timeout -k 10 300 python3 tools/perf_random.py 64 2>&1 | grep -v amdgpu.ids | tee "$out/${round}_random_scene.txt"
timeout -k 10 200 python3 tools/quick_perf_yml.py 128 2>&1 | grep -v amdgpu.ids | tee "$out/${round}_all_scenes.txt"
timeout -k 10 200 python3 tools/c5_check.py 2>&1 | grep -v amdgpu.ids | tee "$out/${round}_c5_check.txt"
timeout -k 10 200 python3 tools/strip_share.py 2>&1 | grep -v amdgpu.ids | tee "$out/${round}_strip_share.txt"
timeout -k 10 120 python3 tools/time_tiles.py 2>&1 | grep -v amdgpu.ids | tee "$out/${round}_tile_stream.txt"
BENCH_REHEARSE_ON_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > "$out/${round}_bench_rehearsal_gpus2.json" 2> "$out/rehearsal.err"
timeout -k 10 300 python3 bench.py --force-dist --steps 5 --warmup 2 --no-cpu-baseline > "$out/${round}_bench_force_dist_rccl.json" 2> "$out/force_dist.err"
python3 -c "
import json
d=json.load(open('$out/${round}_bench_rehearsal_gpus2.json'))
print('rehearsal n_gpus', d['n_gpus'], d['config']['parallelism'], '| frame matches single rank:', d['rehearsal_frame_matches_single_rank'])
d=json.load(open('$out/${round}_bench_force_dist_rccl.json'))
print('force-dist:', d['force_dist'], '|', d['config']['parallelism'], '| %.2f ms | frame matches:' % d['ms_per_step'], d['rehearsal_frame_matches_single_rank'])"

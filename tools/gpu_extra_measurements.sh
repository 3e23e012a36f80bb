#!/bin/bash
# Run on the GPU box: measurements beside the round's main set (tools/gpu_finalize_profiles.sh).
#   tools/gpu_extra_measurements.sh r03
# 1. rocprofv3 --kernel-trace --stats for the C2 / C4 / random bench commands (C3's is in the main set)
# 2. a sustained C3 run (400 timed steps, ~30 s of kernel time): does the rate hold once the card is warm?
# 3. C3 at seeds 1, 2, 3 (the rate must not depend on the seed)
set -o pipefail
round=${1:-r03}
cd "$(dirname "$0")/.."
out=gpurun_out/$round
mkdir -p "$out"
export TMPDIR=/tmp
for w in c2 c4 random; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$out/prof_$w" -o trace -- python3 "$OLDPWD/bench.py" --workload $w --no-cpu-baseline --pmc none --no-host-delivery --steps 10 --warmup 2 > "$OLDPWD/$out/${round}_${w}_bench_under_rocprof.json" 2> "$OLDPWD/$out/rocprof_$w.err") || { tail -5 "$out/rocprof_$w.err"; exit 1; }
  f=$(find "$out/prof_$w" -name '*kernel_stats.csv' | sort | sed -n 1p)
  cp "$f" "$out/${round}_${w}_kernel_stats.csv"
  sed -n 1,3p "$out/${round}_${w}_kernel_stats.csv"
  rm -rf "$out/prof_$w"
done
timeout -k 10 200 python3 bench.py --no-cpu-baseline --pmc none --no-host-delivery --steps 400 --warmup 5 > "$out/${round}_c3_sustained_400_steps.json" 2> "$out/sustained.err" || { tail -5 "$out/sustained.err"; exit 1; }
python3 -c "
import json
d=json.load(open('$out/${round}_c3_sustained_400_steps.json'))
print('sustained: %d steps, %.2f ms per step, %.0f Msamples/s' % (d['steps'], d['ms_per_step'], d['value']))"
for seed in 1 2 3; do
  timeout -k 10 120 python3 bench.py --seed $seed --no-cpu-baseline --pmc none --no-host-delivery --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('seed $seed: %.2f ms per step, %.0f Msamples/s, %.4f segments per sample' % (d['ms_per_step'], d['value'], d['roofline']['segments_per_launch'] / (1920*1080*1024)))"
done | tee "$out/${round}_c3_seeds.txt"

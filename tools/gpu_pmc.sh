#!/bin/bash
# Run on the GPU box (via gpurun): one rocprofv3 PMC pass per counter group
# over a short bench run (no --kernel-trace/--stats together with --pmc: the
# pool refuses that combination).  Results: gpurun_out/pmc_<tag>/g<k>/... and
# gpurun_out/pmc_<tag>/summary.json, which carries the stamp of the kernel
# sources it was collected on (tools/source_stamp.py) — copy it to
# profiles/rNN_<workload>_pmc_summary.json to have bench.py use it.
# Usage: tools/gpu_pmc.sh <tag> "<counters group 1>" ["<counters group 2>" ...] -- [bench args]
#        tools/gpu_pmc.sh <tag> standard -- --workload c4     (the four groups bench.py reads)
set -eo pipefail
tag=$1
shift
groups=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do groups+=("$1"); shift; done
[ "$1" == "--" ] && shift
if [ "${groups[0]}" == "standard" ]; then
    groups=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES"
            "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
            "FETCH_SIZE" "WRITE_SIZE")
fi
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
root=$(pwd)/gpurun_out/pmc_$tag
g=0
for counters in "${groups[@]}"; do
    out=$root/g$g
    mkdir -p "$out"
    # shellcheck disable=SC2086
    rocprofv3 --pmc $counters --output-format csv -d "$out" -o pmc -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > "$out/bench_stdout.log" 2> "$out/rocprof_stderr.log" || { tail -20 "$out/rocprof_stderr.log"; exit 1; }
    f=$(find "$out" -name '*counter_collection.csv' | sort | sed -n 1p)
    echo "== group $g: $counters"
    if [ -n "$f" ]; then
        python3 - "$f" "$root/summary.json" "$out/bench_stdout.log" "$*" <<'PYEOF'
import csv, sys, collections, json, os
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from source_stamp import kernel_source_sha
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary_path = sys.argv[2]
try:
    summary = json.load(open(summary_path))
except Exception:
    summary = {}
bench = None
for line in open(sys.argv[3]):
    if line.startswith("{"):
        bench = json.loads(line)
stamp = summary.setdefault("_stamp", {})
stamp["source_sha"] = kernel_source_sha()
stamp["bench_args"] = sys.argv[4]
if bench:
    stamp["workload"] = bench["config"]["workload"]
    stamp.setdefault("kernel_ms_under_pmc", []).append(bench["roofline"]["kernel_ms"])
    stamp["segments_per_launch"] = bench["roofline"]["segments_per_launch"]
for k, cs in agg.items():
    if "trace" not in k and "resolve" not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
        summary.setdefault(k, {})[c] = {"dispatches": len(v), "mean": sum(v) / len(v)}
json.dump(summary, open(summary_path, "w"), indent=1, sort_keys=True)
PYEOF
    else
        echo "no counter_collection.csv"; ls -R "$out" | head
    fi
    g=$((g + 1))
done

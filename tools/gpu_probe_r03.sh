#!/bin/bash
# Round-3 first GPU call: what counters exist, does RCCL initialise, where do the numbers stand.
set -o pipefail
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
out=gpurun_out/r03_probe
mkdir -p $out
echo "== list-avail"; (cd /tmp && rocprofv3 --list-avail > $OLDPWD/$out/list_avail.txt 2>&1); grep -c . $out/list_avail.txt
grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_INST_[A-Z0-9_]*VALU[A-Z0-9_]*\|SQ_ACTIVE_INST_[A-Z_0-9]*\|SQ_VALU_[A-Z0-9_]*\|SQ_INSTS_[A-Z0-9_]*" $out/list_avail.txt | sort -u | tr '\n' ' '; echo
echo "== host-visible stores"; timeout -k 5 60 ./build/host_visible | tee $out/host_visible.txt
echo "== RCCL world 1"; timeout -k 10 300 python3 bench.py --force-dist --steps 3 --warmup 1 --no-cpu-baseline > $out/force_dist.json 2> $out/force_dist.err; echo "rc $?"; tail -3 $out/force_dist.err; cut -c1-400 $out/force_dist.json
echo "== baseline"; tools/perf_ab.sh base | tee $out/base.txt
echo "== fence probes (dev build)"
tools/perf_ab.sh dev RT_DBG0=0 | tee -a $out/base.txt
tools/perf_ab.sh fence_agent RT_DBG3=77 | tee -a $out/base.txt
tools/perf_ab.sh fence_system RT_DBG3=78 | tee -a $out/base.txt
echo "== random"; timeout -k 10 200 python3 tools/perf_random.py 64 bvh | tee $out/random.txt

#!/bin/bash
# A/B of the library in the tree against build/prev.so (a copy of an earlier build): are the frames bit-identical,
# and what do the three BASELINE workloads (+ WORKLOADS=...) cost with each?
cd "$(dirname "$0")/.."
timeout -k 10 200 python3 tools/compare_builds.py build/prev.so 2>&1 | grep -v amdgpu.ids
RACER_TRACER_AMD_LIB=$PWD/build/prev.so tools/perf_ab.sh prev
tools/perf_ab.sh new

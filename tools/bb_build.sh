#!/bin/bash
# Builds racer-tracer_amd/build/libracer_tracer_amd_bb_<variant>.so: the product library with basic-block counters in ONE
# variant of k_trace_pool_f64 (tools/bb_instrument.py), plus build/bb/blocks_<variant>.json.  Run here (no GPU needed);
# tools/bb_profile.py runs it on the GPU box.
#   tools/bb_build.sh Li0ELb0ELb0ELb0E [more variants ...]      (PRIMS, TEXTURED, SPECULAR, BVH of the template)
set -eo pipefail
cd "$(dirname "$0")/../racer-tracer_amd"
LLVM=/opt/rocm/lib/llvm/bin
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRT_BB_COUNT"
make -j8 lib/libracer_tracer_amd.so >/dev/null     # the other objects are the product's own
mkdir -p build/bb
$HIPCC $FLAGS --cuda-device-only -S csrc/rt_trace_pool_kernel.hip -o build/bb/dev.s 2>/dev/null
$HIPCC $FLAGS -gline-tables-only --cuda-device-only -S csrc/rt_trace_pool_kernel.hip -o build/bb/dev_g.s 2>/dev/null
for v in "$@"; do
  python3 ../tools/bb_instrument.py build/bb/dev.s build/bb/dev_g.s "$v" build/bb/dev_$v.s build/bb/blocks_$v.json
  $LLVM/clang -target amdgcn-amd-amdhsa -mcpu=gfx950 -c build/bb/dev_$v.s -o build/bb/dev_$v.o
  $LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared build/bb/dev_$v.o -o build/bb/dev_$v.out
  $LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
      -input=/dev/null -input=build/bb/dev_$v.out -output=build/bb/dev_$v.hipfb
  $HIPCC $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang build/bb/dev_$v.hipfb -c csrc/rt_trace_pool_kernel.hip -o build/bb/rt_trace_pool_kernel_$v.o 2>/dev/null
  objs=$(ls build/product/*.o | grep -v "/rt_trace_pool_kernel.o")
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o build/libracer_tracer_amd_bb_$v.so $objs build/bb/rt_trace_pool_kernel_$v.o -lz -lpthread
  echo "built build/libracer_tracer_amd_bb_$v.so"
done

#!/usr/bin/env python3
"""Static instruction counts per region of k_trace_pool_f64 from `make -C racer-tracer_amd isa`
(the listing carries the RT_REGION boundaries as comments).  Layout order, not execution
order: a count says how much code sits between two markers, loops counted once.

usage: tools/isa_regions.py [variant substring, default Li0ELb0ELb0ELb0E]"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LISTING = os.path.join(ROOT, "racer-tracer_amd", "build", "isa", "rt_trace_pool_kernel-hip-amdgcn-amd-amdhsa-gfx950.s")
NAMES = {0: "item setup", 1: "batches", 2: "hand-out + primary ray", 3: "closest hit", 4: "miss / material", 5: "sampler",
         6: "scatter + accumulate", 7: "item end", 8: "hit record", 9: "texture, step 1", 10: "Noise rounds"}


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "Li0ELb0ELb0ELb0E"
    lines = open(LISTING).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN10rtdev_fast16k_trace_pool_f64I" + want))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    seg = collections.Counter()
    kinds = collections.defaultdict(collections.Counter)
    cur = collections.Counter()
    order = []
    for l in lines[start:end]:
        m = re.search(r"==== end of region (\d+)", l)
        if m:
            k = int(m.group(1))
            order.append((k, cur))
            cur = collections.Counter()
            continue
        m = re.match(r"\s+([a-z][a-z0-9_]+)\s", l + " ")
        if m and not l.strip().startswith((";", ".")):
            op = m.group(1)
            cls = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "mem"
            cur[cls] += 1
            cur["op:" + op] += 1
    print("variant", want, "(code between consecutive markers, in layout order)")
    for k, c in order:
        top = sorted(((v, o[3:]) for o, v in c.items() if o.startswith("op:v_") or o.startswith("op:ds_")), reverse=True)[:8]
        print("-> %-24s valu %4d  salu %4d  lds %3d  mem %3d   %s" % (NAMES.get(k, str(k)), c["valu"], c["salu"], c["lds"], c["mem"],
                                                                  " ".join("%s:%d" % (o, v) for v, o in top)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Rewrites the three workload rows of DESIGN.md's round table from profiles/r02_<w>_bench.json, so that the
document quotes exactly what is committed under profiles/.  Run after tools/gpu_finalize_profiles.sh."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"c3": ("C3 cornell_box 1080p×1024", "20 365"), "c2": ("C2 three_balls 1080p×256", "27 204"),
         "c4": ("C4 noise_and_textures 1080p×512", "13 886")}


def row(w):
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_%s_bench.json" % w)))
    r = d["roofline"]
    name, round1 = NAMES[w]
    return "| %s | %s | %.1f | %.1f | %.3f | %.1f | %.2f | %.0f GB/s | %s |" % (
        name, format(int(round(d["value"])), ",").replace(",", " "), d["ms_per_step"], r["gsegments_per_s"], r["frac"],
        r["lanes_per_inst"], r["valu_insts_per_segment"], r["traffic"], round1)


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    a = s.index("| C3 cornell_box 1080p×1024 |")
    b = s.index("\n\nAll three launches are VALU-issue bound")
    s = s[:a] + "\n".join(row(w) for w in ("c3", "c2", "c4")) + s[b:]
    open(path, "w").write(s)
    print("\n".join(row(w) for w in ("c3", "c2", "c4")))


if __name__ == "__main__":
    main()

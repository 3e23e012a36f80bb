#!/usr/bin/env python3
"""Rewrites the workload rows of DESIGN.md's round table from profiles/<round>_<w>_bench.json, so that the
document quotes exactly what is committed under profiles/.  Run after tools/gpu_finalize_profiles.sh."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = "r03"
NAMES = {"c3": ("C3 cornell_box 1080p×1024", "25 934"), "c2": ("C2 three_balls 1080p×256", "32 373"),
         "c4": ("C4 noise_and_textures 1080p×512", "19 171")}


def row(w):
    d = json.load(open(os.path.join(ROOT, "profiles", "%s_%s_bench.json" % (ROUND, w))))
    r, h = d["roofline"], d["host_delivered"]
    name, before = NAMES[w]
    return "| %s | %s | %.1f | %.1f | %.3f | %.2f | %.2f | %.1f | %.2f | %.0f GB/s | %.1f / %.1f | %s |" % (
        name, format(int(round(d["value"])), ",").replace(",", " "), d["ms_per_step"], r["gsegments_per_s"], r["frac"],
        r["issue_frac"], r["valu_busy"], r["lanes_per_inst"], r["valu_insts_per_segment"], r["traffic"],
        h["rt_render"]["ms"], h["rt_render_frame"]["ms"], before)


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    a = s.index("| C3 cornell_box 1080p×1024 |")
    b = s.index("\n\nAll three launches")
    s = s[:a] + "\n".join(row(w) for w in ("c3", "c2", "c4")) + s[b:]
    open(path, "w").write(s)
    print("\n".join(row(w) for w in ("c3", "c2", "c4")))


if __name__ == "__main__":
    main()

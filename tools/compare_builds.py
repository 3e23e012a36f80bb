#!/usr/bin/env python3
"""Developer probe: do two builds of the library render bit-identical frames?
usage: tools/compare_builds.py <other .so> [spp] [fast|reference] [max |diff| that still counts as equal, default 0]
(the default build is the other side)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd")
host = importlib.import_module("racer-tracer_amd.host")

other = rt.load_library(sys.argv[1])
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
arith = rt.abi.RT_ARITH_REFERENCE if len(sys.argv) > 3 and sys.argv[3] == "reference" else rt.abi.RT_ARITH_FAST
tol = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
bad = 0
for scene_name in ("cornell_box.yml", "three_balls.yml", "noise_and_textures.yml", "emissive.yml", "clown.yml", "cornell_box_boxes.yml", "random"):
    path = scene_name if scene_name == "random" else os.path.join(ROOT, "scenes", scene_name)
    s = host.Session(os.path.join(ROOT, "scenes", "config_c2.yml"), scene=path)
    p = s.params
    p.width, p.height, p.samples = 480, 270, spp
    frames = []
    for lib in (None, other):
        sc = rt.Scene(s, library=lib, arithmetic=arith)
        frames.append(sc.render_frame(s.camera, p))
        sc.close()
    same = np.array_equal(frames[0], frames[1])
    bad += not same and not float(np.abs(frames[0] - frames[1]).max()) <= tol
    differ = int((np.abs(frames[0] - frames[1]).max(axis=-1) > 0).sum())
    print("%-24s %s  max |diff| %.3g  (%d of %d pixels differ)" % (scene_name, "bit-identical" if same else "DIFFERENT", float(np.abs(frames[0] - frames[1]).max()),
                                                                  differ, p.width * p.height))
sys.exit(1 if bad else 0)

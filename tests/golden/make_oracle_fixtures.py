#!/usr/bin/env python3
"""Generates tests/golden/oracle_frames.npz: small float frames rendered by
the CPU oracle (oracle/liboracle.so) at seed 1, one per shipped scene, through
the C++ scene loader's POD output.  The GPU parity tests compare the HIP path
against these committed vectors as well as against the live oracle, so a
silent change of the RNG contract or of either implementation shows up as a
diff against history.

Each entry: <scene>_frame (float64 [H, W, 3], gamma-encoded, not tone-mapped),
<scene>_segments, and the render settings.  Re-run after any deliberate change
of include/rt_rng.h or of the oracle:  python tests/golden/make_oracle_fixtures.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

SCENES = {  # name: (config, width, height, spp, use_bvh)
    "three_balls": ("config_c1.yml", 64, 36, 8, 1),
    "cornell_box": ("config_c3.yml", 64, 36, 8, 1),
    "noise_and_textures": ("config_c4.yml", 64, 36, 8, 1),
    "cornell_box_boxes": ("config_c3.yml", 64, 36, 8, 0),
    "emissive": ("config_c3.yml", 64, 36, 8, 1),
    "clown": ("config_c3.yml", 64, 36, 8, 1),
    "two_balls": ("config_c1.yml", 64, 36, 8, 1),
    # the procedural scene of scene/random.rs (485 spheres, 383 of them moving) at seed 1
    "random": ("config_c1.yml", 64, 36, 8, 1),
}


def load(host, name):
    cfg, w, h, spp, use_bvh = SCENES[name]
    scene = "random" if name == "random" else os.path.join(ROOT, "scenes", name + ".yml")
    s = host.Session(os.path.join(ROOT, "scenes", cfg), scene=scene)
    p = s.params
    p.width, p.height, p.samples = w, h, spp
    return s, p, use_bvh


def camera_for(host, session, p):
    """Re-derive the camera for the fixture's aspect ratio from the session's own camera."""
    c = session.camera
    origin = tuple(c.origin)
    look_at = tuple(c.origin[k] - c.forward[k] for k in range(3))
    return host.camera_new(origin, look_at, c.vfov, c.lens_radius * 2.0, c.focus_distance, p.width, p.height)


def main():
    host = importlib.import_module("racer-tracer_amd.host")
    from oracle import oracle_ctypes as orc
    out = {}
    for name in SCENES:
        s, p, use_bvh = load(host, name)
        cam = camera_for(host, s, p)
        frame, segs = orc.render(s.desc, cam, p, use_bvh=use_bvh)
        out[name + "_frame"] = frame
        out[name + "_segments"] = np.int64(segs)
        print("%-20s segments/sample %.3f  mean %s" % (name, segs / (p.width * p.height * p.samples),
                                                       np.round(frame.mean(axis=(0, 1)), 4)))
    np.savez_compressed(os.path.join(HERE, "oracle_frames.npz"), **out)
    print("wrote oracle_frames.npz")


if __name__ == "__main__":
    main()

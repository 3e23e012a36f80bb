#!/usr/bin/env python3
"""Times rt_render's progressive tile stream against rt_render_frame on the C3
frame (cornell_box 1920x1080x1024) and prints when each tile column arrived."""
import ctypes as C
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd")
host = importlib.import_module("racer-tracer_amd.host")
abi = importlib.import_module("racer-tracer_amd.abi")

session = host.Session(os.path.join(ROOT, "scenes", "config_c3.yml"), scene=os.path.join(ROOT, "scenes", "cornell_box.yml"))
p = session.params
library = rt.load_library(os.environ["RT_LIB"]) if os.environ.get("RT_LIB") else None  # e.g. the dev build
scene = rt.Scene(session, device=0, library=library)
scene.render_frame(session.camera, p)
t0 = time.time()
scene.render_frame(session.camera, p)
t_frame = time.time() - t0
arrivals = []


def on_tile(_user, rgb, r, c, w, h):
    arrivals.append((time.time(), r, c))


cb = abi.RtTileCallback(on_tile)
for label, cancel in (("no cancel flag", None), ("with cancel flag", C.pointer(C.c_int(0)))) * 2:
    del arrivals[:]
    t0 = time.time()
    rc = scene._lib.rt_render(scene._h, C.byref(session.camera), C.byref(p), cb, None, cancel)
    t_tiles = time.time() - t0
    st = scene.last_stats()
    cols = sorted({c for _, _, c in arrivals})
    first = {c: min(t for t, _, cc in arrivals if cc == c) - t0 for c in cols}
    print("%s: rc=%d rt_render %.1f ms (%d tiles, %d launches) vs rt_render_frame incl. 50 MB copy %.1f ms"
          % (label, rc, t_tiles * 1e3, len(arrivals), st.kernel_launches, t_frame * 1e3))
    print("  first tile of each column at ms:", " ".join("%.0f" % (first[c] * 1e3) for c in cols))
scene.close()

"""Developer probe: kernel time of the trace kernel on the hand-built scenes.

Usage: python tools/quick_perf.py [spp] [scene ...]
Prints Msamples/s and Gsegments/s from the library's own HIP-event timing.
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("racer-tracer_amd")
import scenes_py as S  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
names = sys.argv[2:] or ["cornell_box", "three_balls", "cornell_box_boxes"]
W, H = 1920, 1080
for name in names:
    bundle, cam, _ = getattr(S, name)()
    camera = S.camera_for(cam, W, H)
    params = S.abi.render_params(W, H, spp)
    scene = rt.Scene(bundle)
    scene.render_frame(camera, S.abi.render_params(W, H, 1))  # warm-up
    t0 = time.time()
    scene.render_frame(camera, params)
    wall = time.time() - t0
    st = scene.last_stats()
    ms = st.kernel_ms
    print("%-18s spp=%d kernel %.2f ms (wall %.1f ms incl. D2H) | %.1f Msamples/s | %.2f Gseg/s | %.2f seg/sample"
          % (name, spp, ms, wall * 1e3, st.samples / ms / 1e3, st.segments / ms / 1e6,
             st.segments / st.samples), flush=True)
    scene.close()

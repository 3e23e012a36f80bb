"""Developer probe: kernel time of the C3 frame against samples per pixel — the intercept of the line is what
one launch costs beyond its share of the work (it is what limits strong scaling over GPUs).
Usage: python tools/launch_fit.py [scene.yml]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd")
host = importlib.import_module("racer-tracer_amd.host")
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box.yml"
s = host.Session(os.path.join(ROOT, "scenes", "config_c3.yml"), scene=os.path.join(ROOT, "scenes", name))
p = s.params
scene = rt.Scene(s)
xs, ys = [], []
for spp in (64, 128, 256, 384, 512, 768, 1024, 2048):
    p.samples = spp
    best = 1e9
    for _ in range(3):
        scene.render_frame(s.camera, p)
        best = min(best, scene.last_stats().kernel_ms)
    xs.append(spp); ys.append(best)
    print("spp %5d  kernel %8.3f ms  %.4f ms per spp" % (spp, best, best / spp), flush=True)
b, a = np.polyfit(xs[2:], ys[2:], 1)
print("fit over spp >= 256: %.3f ms + %.5f ms/spp" % (a, b))
scene.close()

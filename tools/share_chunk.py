import importlib, os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd")
host = importlib.import_module("racer-tracer_amd.host")
s = host.Session(os.path.join(ROOT, "scenes", "config_c3.yml"), scene=os.path.join(ROOT, "scenes", "cornell_box.yml"))
p = s.params
scene = rt.Scene(s)
scene.render_frame(s.camera, p)
out = []
for n in (1, 8, 16, 32):
    p.strip_rows, p.strip_count, p.strip_index = 8, n, 0
    best = 1e9
    for _ in range(3):
        scene.render_frame(s.camera, p)
        best = min(best, scene.last_stats().kernel_ms)
    out.append("N=%d %.2f" % (n, best))
print("chunk", os.environ.get("RT_POOL_CHUNK", "default"), "taper", "off" if os.environ.get("RT_POOL_NO_TAPER") else "on", " ".join(out), flush=True)
scene.close()

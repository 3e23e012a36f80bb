"""Developer probe: how long does one rank's share of the C3 frame take for N = 1, 2, 4, 8
(kernel + resolve, HIP events), against the ideal full/N?  Run on one GPU."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd")
host = importlib.import_module("racer-tracer_amd.host")
s = host.Session(os.path.join(ROOT, "scenes", "config_c3.yml"), scene=os.path.join(ROOT, "scenes", "cornell_box.yml"))
p = s.params
scene = rt.Scene(s)
scene.render_frame(s.camera, p)
full = None
for n in (1, 2, 4, 8):
    worst = 0.0
    times = []
    for rank in range(n):
        p.strip_rows, p.strip_count, p.strip_index = 8, n, rank
        scene.render_frame(s.camera, p)
        st = scene.last_stats()
        worst = max(worst, st.kernel_ms + st.resolve_ms)
        times.append("%.2f+%.2f" % (st.kernel_ms, st.resolve_ms))
    if n == 1:
        full = worst
    print("N=%d slowest rank %.2f ms  ideal %.2f ms  efficiency %.1f %%" % (n, worst, full / n, 100.0 * full / n / worst), " ".join(times), flush=True)
scene.close()

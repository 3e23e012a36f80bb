"""Developer probe: what does a 1/8 share of the C3 frame cost beyond an eighth of the frame, and does that excess grow with
the work (imbalance at the end of the launch) or stay (a fixed cost per launch)?  Two-pass path, HIP events."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd"); host = importlib.import_module("racer-tracer_amd.host")
s = host.Session(os.path.join(ROOT, "scenes", "config_c3.yml"), scene=os.path.join(ROOT, "scenes", "cornell_box.yml"))
p = s.params
scene = rt.Scene(s)
frame = torch.zeros((p.height, p.width, 3), dtype=torch.float64, device="cuda")
def two_pass():
    scene.render_frame_device(s.camera, p, frame.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    st = scene.last_stats(); return st.kernel_ms
for spp in ([int(x) for x in os.environ["SPPS"].split(",")] if "SPPS" in os.environ else (128, 512, 1024, 4096)):
    p.samples = spp
    t = {}
    for n in (1, 8):
        p.strip_rows, p.strip_count, p.strip_index = (8, n, 3 % n)
        two_pass()
        t[n] = min(two_pass() for _ in range(3))
    print("spp %5d: full %.2f ms, 1/8 share %.3f ms, excess over an eighth %.3f ms (%.1f %%)" % (spp, t[1], t[8], t[8] - t[1] / 8, 100 * (t[8] / (t[1] / 8) - 1)), flush=True)

import importlib, os, sys, time
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd")
host = importlib.import_module("racer-tracer_amd.host")
import torch
s = host.Session(os.path.join(ROOT, "scenes", "config_c3.yml"), scene=os.path.join(ROOT, "scenes", "cornell_box.yml"))
p = s.params
p.width, p.height, p.samples = 3840, 2160, 4096
cam = host.camera_new(tuple(s.camera.origin), tuple(s.camera.origin[k] - s.camera.forward[k] for k in range(3)), s.camera.vfov, 0.0, s.camera.focus_distance, p.width, p.height)
sc = rt.Scene(s)
frame = torch.zeros((p.height, p.width, 3), dtype=torch.float64, device="cuda")
for share in (8, 1):
    p.strip_rows, p.strip_count, p.strip_index = 8, share, 0
    t0 = time.time()
    sc.render_frame_device(cam, p, frame.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = sc.last_stats()
    print("C5 3840x2160x4096 share 1/%d: %.1f ms kernel, %.1f Msamples/s, %.2f seg/sample, finite %s" % (share, st.kernel_ms, st.samples / st.kernel_ms / 1e3, st.segments / st.samples, bool(torch.isfinite(frame).all())), flush=True)
sc.close()

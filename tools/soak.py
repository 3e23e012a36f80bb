"""Developer probe: 25 renders of each BASELINE scene at 1080p x 128 spp must give ONE frame (sha256):
the queue order is timing dependent, the result must not be."""
import importlib, os, sys, hashlib
import torch
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd"); host = importlib.import_module("racer-tracer_amd.host")
for wl, cfg, scene in (("c3", "config_c3.yml", "cornell_box.yml"), ("c4", "config_c4.yml", "noise_and_textures.yml"), ("c2", "config_c2.yml", "three_balls.yml")):
    s = host.Session(os.path.join(ROOT, "scenes", cfg), scene=os.path.join(ROOT, "scenes", scene))
    p = s.params
    p.samples = 128
    sc = rt.Scene(s)
    frame = torch.zeros((p.height, p.width, 3), dtype=torch.float64, device="cuda")
    hashes = set()
    for i in range(25):
        sc.render_frame_device(s.camera, p, frame.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        hashes.add(hashlib.sha256(frame.cpu().numpy().tobytes()).hexdigest())
    print(wl, "25 renders ->", len(hashes), "distinct frame(s)", flush=True)
    sc.close()

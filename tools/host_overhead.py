"""two-pass kernel+resolve vs rt_render_frame / rt_render wall time on several workloads"""
import importlib, os, sys, time, ctypes as C
import numpy as np
ROOT="/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, ROOT)
import torch
rt = importlib.import_module("racer-tracer_amd"); host = importlib.import_module("racer-tracer_amd.host")
import bench
for name, spp in (("c3",0),("c2",0),("c4",0),("random",64),("random",256),("c3",64)):
    session, wl = bench.load_workload(host, name, spp)
    p = session.params
    scene = rt.Scene(session)
    out = torch.zeros((p.height,p.width,3), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def two_pass():
        scene.render_frame_device(session.camera, p, out.data_ptr(), st); torch.cuda.synchronize()
    two_pass()
    t=[]; 
    for _ in range(3):
        t0=time.perf_counter(); two_pass(); t.append(time.perf_counter()-t0)
    base=min(t)*1e3
    hd = bench.host_delivery(rt, scene, session, reps=3)
    print("%-45s two-pass %.2f ms | rt_render_frame %.2f (+%.2f) | rt_render %.2f (+%.2f)" % (wl, base, hd["rt_render_frame"]["ms"], hd["rt_render_frame"]["ms"]-base, hd["rt_render"]["ms"], hd["rt_render"]["ms"]-base), flush=True)
    scene.close()

import importlib, os, sys, time
import numpy as np, torch
ROOT=os.getcwd(); sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd"); host = importlib.import_module("racer-tracer_amd.host")
s = host.Session(os.path.join(ROOT,"scenes","config_c5.yml"), scene=os.path.join(ROOT,"scenes","cornell_box.yml"))
p = s.params; p.samples = 128
sc = rt.Scene(s)
dev = torch.zeros((p.height,p.width,3), dtype=torch.float64, device="cuda")
sc.render_frame_device(s.camera, p, dev.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
want = dev.cpu().numpy()
t0=time.time(); got = sc.render_frame(s.camera, p); t1=time.time()
tiles = sc.render_tiles(s.camera, p); t2=time.time()
ok = np.array_equal(got, want) and all(np.array_equal(a, want[r:r+h, c:c+w]) for r,c,w,h,a in tiles)
print("C5 frame %dx%d x %d spp: rt_render_frame %.1f ms, rt_render %.1f ms (%d tiles), delivered == two-pass: %s" % (p.width,p.height,p.samples,(t1-t0)*1e3,(t2-t1)*1e3,len(tiles),ok))
sc.close()

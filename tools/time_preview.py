"""Developer probe: what the reference's interactive mode meets — the PREVIEW renderer (RtRenderParams.scale > 1,
renderer/cpu_scaled.rs) through rt_render's tile stream and rt_render_frame at 1920x1080."""
import ctypes as C, importlib, os, sys, time
import numpy as np
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd"); host = importlib.import_module("racer-tracer_amd.host")
abi = rt.abi
s = host.Session(os.path.join(ROOT, "scenes", "config_c3.yml"), scene=os.path.join(ROOT, "scenes", "cornell_box.yml"))
p = s.preview_params
print("preview params: %dx%d, %d spp, depth %d, scale %d, tiles %dx%d" % (p.width, p.height, p.samples, p.max_depth, p.scale, p.tiles_w, p.tiles_h))
sc = rt.Scene(s)
frame = np.zeros((p.height, p.width, 3))
n = [0]
def on_tile(_u, rgb, r, c, w, h):
    n[0] += 1
cb = abi.RtTileCallback(on_tile)
lib = rt.lib()
for name, call in (("rt_render (tile stream)", lambda: lib.rt_render(sc._h, C.byref(s.camera), C.byref(p), cb, None, None)),
                   ("rt_render_frame", lambda: lib.rt_render_frame(sc._h, C.byref(s.camera), C.byref(p), frame.ctypes.data_as(C.POINTER(C.c_double))))):
    call()
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); rc = call(); t.append(time.perf_counter() - t0)
    st = sc.last_stats()
    print("%-26s rc %d  %.2f ms per call (kernel %.2f + resolve %.2f ms on the device)" % (name, rc, min(t) * 1e3, st.kernel_ms, st.resolve_ms))
sc.close()

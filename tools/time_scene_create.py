"""Developer probe: rt_scene_create (upload + BVH build) — the reference rebuilds its BVH between renders (main.rs:178)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.getcwd(); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes_py as S
import test_gpu_parity_proofs as T
rt = importlib.import_module("racer-tracer_amd"); host = importlib.import_module("racer-tracer_amd.host")
cases = [("cornell_box (6 rects)", S.cornell_box()[0]), ("random (485 spheres)", host.Session(os.path.join(ROOT, "scenes", "config_c2.yml"), scene="random")),
         ("hall of 3 000 spheres", T.hall_of_spheres(3000, False)[0]), ("hall of 20 000 spheres", T.hall_of_spheres(20000, False)[0])]
rt.Scene(cases[0][1]).close()
for name, desc in cases:
    t = []
    for _ in range(3):
        t0 = time.perf_counter(); sc = rt.Scene(desc); t.append(time.perf_counter() - t0); sc.close()
    print("%-26s rt_scene_create %.2f ms" % (name, min(t) * 1e3), flush=True)

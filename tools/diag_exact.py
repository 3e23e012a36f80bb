"""Developer probe: the 20 000-sphere hall of mirrors through both arithmetics (RtSceneOptions.arithmetic) vs the oracle's two closest-hit routines."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes_py as S
import test_gpu_parity_proofs as T
from oracle import oracle_ctypes as orc
rt = importlib.import_module("racer-tracer_amd")
for n in (3000, 20000):
    bundle, cam = T.hall_of_spheres(n, False)
    w, h, spp = 96, 54, 3
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp, max_depth=8)
    refs = {b: orc.render(bundle.desc, camera, params, use_bvh=b) for b in (0, 1)}
    print(n, "oracle linear vs oracle bvh: segs", refs[0][1], refs[1][1], "max diff", np.abs(refs[0][0] - refs[1][0]).max(), "pixels >1e-9:", int((np.abs(refs[0][0] - refs[1][0]).max(axis=-1) > 1e-9).sum()))
    for name, arith, hit in (("fast bvh", S.abi.RT_ARITH_FAST, S.abi.RT_HIT_BVH), ("reference-arithmetic bvh", S.abi.RT_ARITH_REFERENCE, S.abi.RT_HIT_BVH)):
        sc = rt.Scene(bundle, arithmetic=arith, closest_hit=hit)
        got = sc.render_frame(camera, params); st = sc.last_stats(); sc.close()
        for b in (0, 1):
            d = np.abs(got - refs[b][0])
            print("  %-12s vs oracle use_bvh=%d: segs %d vs %d  max %.3g  pixels>1e-9 %d  pixels>1e-3 %d  median %.3g"
                  % (name, b, st.segments, refs[b][1], d.max(), int((d.max(axis=-1) > 1e-9).sum()), int((d.max(axis=-1) > 1e-3).sum()), np.median(d)))

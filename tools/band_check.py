"""Developer probe: the default scene (cornell_box_boxes) at 1080p x 128 spp against the oracle on bands of rows, through
the oracle's BVH and its linear scan.  usage: tools/band_check.py [other .so]"""
import importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd"); host = importlib.import_module("racer-tracer_amd.host")
from oracle import oracle_ctypes as orc
s = host.Session(os.path.join(ROOT, "scenes", "config_c3.yml"), scene=os.path.join(ROOT, "scenes", "cornell_box_boxes.yml"))
p = s.params; p.samples = int(os.environ.get("SPP", "128"))
sc = rt.Scene(s)
got = sc.render_frame(s.camera, p); sc.close()
p.strip_rows, p.strip_count, p.strip_index = 2, 180, 100
rows = ((np.arange(p.height) // 2) % 180) == 100
idx = np.nonzero(rows)[0]
for use_bvh in (1, 0):
    ref, _ = orc.render(s.desc, s.camera, p, use_bvh=use_bvh)
    d = np.abs(ref[rows] - got[rows]).max(axis=-1)
    print("oracle use_bvh=%d: max %.3g, pixels > 1e-3: %d, > 1e-9: %d of %d" % (use_bvh, d.max(), int((d > 1e-3).sum()), int((d > 1e-9).sum()), d.size))
    for k, r in enumerate(idx):
        print("   row %4d: > 1e-9: %4d   > 1e-3: %3d   columns %s" % (r, int((d[k] > 1e-9).sum()), int((d[k] > 1e-3).sum()), np.nonzero(d[k] > 1e-9)[0][[0, -1]].tolist() if (d[k] > 1e-9).any() else []))

"""Developer probe: where does the BVH walk overtake the linear closest-hit loop?  (rt_api.hip: kBvhThreshold)
Scenes of N spheres (Lambertian / Metal / Dielectric mix on a ground sphere), 1920x1080 x 32 spp, both closest-hit routines."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.getcwd(); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes_py as S
rt = importlib.import_module("racer-tracer_amd")
abi = S.abi


def scene(n, seed=7):
    rng = np.random.default_rng(seed)
    textures = [abi.solid((0.5, 0.5, 0.5)), abi.solid((0.8, 0.3, 0.3)), abi.solid((0.3, 0.8, 0.3)), abi.solid((0.9, 0.9, 0.9))]
    materials = [abi.material(abi.RT_MAT_LAMBERTIAN, 0), abi.material(abi.RT_MAT_LAMBERTIAN, 1), abi.material(abi.RT_MAT_LAMBERTIAN, 2),
                 abi.material(abi.RT_MAT_METAL, 3, fuzz=0.1), abi.material(abi.RT_MAT_DIELECTRIC, -1, ior=1.5)]
    prims = [abi.sphere((0.0, -1000.0, 0.0), 1000.0, 0, 1)]
    side = int(np.ceil(np.sqrt(n - 1)))
    for k in range(n - 1):
        x, z = (k % side) - side / 2 + 0.8 * rng.random(), (k // side) - side / 2 + 0.8 * rng.random()
        prims.append(abi.sphere((float(x) * 1.2, 0.3, float(z) * 1.2), 0.3, 1 + int(k % 4), k + 2))
    return abi.SceneBundle(prims, materials, textures, abi.sky())


w, h, spp = 1920, 1080, 32
cam = S.camera_for(dict(look_from=(0.0, 4.0, 14.0), look_at=(0.0, 0.3, 0.0), vfov=35.0, aperture=0.0, focus_distance=10.0), w, h)
p = abi.render_params(w, h, spp)
out = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
for n in (8, 16, 24, 32, 40, 48, 64, 96, 128):
    b = scene(n)
    ms = {}
    for name, mode in (("linear", abi.RT_HIT_LINEAR), ("bvh", abi.RT_HIT_BVH)):
        sc = rt.Scene(b, closest_hit=mode)
        for _ in range(2):
            sc.render_frame_device(cam, p, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        ms[name] = sc.last_stats().kernel_ms
        sc.close()
    print("%4d spheres: linear %7.2f ms   bvh %7.2f ms   -> %s" % (n, ms["linear"], ms["bvh"], "bvh" if ms["bvh"] < ms["linear"] else "linear"), flush=True)

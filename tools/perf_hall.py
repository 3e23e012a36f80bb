"""Developer probe: the BVH variants on LARGE scenes — a hall of N spheres (Lambertian / Metal / Dielectric mixed), 1080p.
Above 1023 nodes the node array (32 B per node) no longer fits the 32 KiB of LDS the walk stages it in and every descent
step is two 128-bit reads from global memory instead (racer-tracer_amd/csrc/rt_api.hip: bvh_nodes_in_lds).
Usage: python3 tools/perf_hall.py [spp] [N ...]        (RACER_TRACER_AMD_LIB=.../libracer_tracer_amd_regions.so adds
the walk's nodes / leaf primitives per segment)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes_py as S
import test_gpu_parity_proofs as T
rt = importlib.import_module("racer-tracer_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sizes = [int(a) for a in sys.argv[2:]] or [300, 1000, 3000, 20000]
w, h = 1920, 1080
for n in sizes:
    bundle, cam = T.hall_of_spheres(n, False)
    camera = S.camera_for(cam, w, h)
    t0 = time.perf_counter()
    sc = rt.Scene(bundle)
    t_create = (time.perf_counter() - t0) * 1e3
    p = S.abi.render_params(w, h, spp)
    sc.render_frame(camera, S.abi.render_params(w, h, 1))
    sc.render_frame(camera, p)
    st = sc.last_stats()
    sc.close()
    print("hall of %5d spheres, 1080p x %d spp: kernel %7.2f ms | %7.1f Msamples/s | %5.2f G segments/s | %.2f segments/sample | rt_scene_create %.1f ms | nodes %s"
          % (n, spp, st.kernel_ms, st.samples / st.kernel_ms / 1e3, st.segments / st.kernel_ms / 1e6, st.segments / st.samples, t_create,
             "in LDS" if n * 32 * 2 // 3 < 32 * 1024 else "in global memory"), flush=True)

"""Developer probe (regions build): region shares, lanes and BVH walk statistics of the `random` scene.
Usage: RACER_TRACER_AMD_LIB=racer-tracer_amd/build/libracer_tracer_amd_regions.so python tools/random_regions.py [spp]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd"); host = importlib.import_module("racer-tracer_amd.host")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = host.Session(os.path.join(ROOT, "scenes", "config_c2.yml"), scene="random")
p = s.params; p.samples = spp
sc = rt.Scene(s); sc.render_frame(s.camera, p); st = sc.last_stats(); sc.close()
print("random 1080p %dspp: kernel %.1f ms %.2f Gseg/s" % (spp, st.kernel_ms, st.segments / st.kernel_ms / 1e6))

"""Developer probe (regions build): when do the waves of ONE rank's share of the C3 frame start and end?
Usage: RACER_TRACER_AMD_LIB=racer-tracer_amd/build/libracer_tracer_amd_regions.so python tools/share_waves.py [N]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd")
host = importlib.import_module("racer-tracer_amd.host")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = host.Session(os.path.join(ROOT, "scenes", "config_c3.yml"), scene=os.path.join(ROOT, "scenes", "cornell_box.yml"))
p = s.params
scene = rt.Scene(s)
scene.render_frame(s.camera, p)
p.strip_rows, p.strip_count, p.strip_index = 8, n, 0
for _ in range(2):
    scene.render_frame(s.camera, p)
    st = scene.last_stats()   # the regions build prints its report here
    print("share 1/%d: kernel %.2f ms" % (n, st.kernel_ms), flush=True)
scene.close()

"""Developer probe: kernel time on the shipped scene files through the host loader.
Usage: python tools/quick_perf_yml.py [spp] [scene ...]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd")
host = importlib.import_module("racer-tracer_amd.host")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
names = sys.argv[2:] or ["cornell_box", "three_balls", "noise_and_textures", "emissive", "clown", "cornell_box_boxes"]
for name in names:
    s = host.Session(os.path.join(ROOT, "scenes", "config_c3.yml"), scene=os.path.join(ROOT, "scenes", name + ".yml"))
    p = s.params
    p.samples = spp
    scene = rt.Scene(s)
    p1 = rt.abi.RtRenderParams.from_buffer_copy(p)
    p1.samples = 1
    scene.render_frame(s.camera, p1)
    scene.render_frame(s.camera, p)
    st = scene.last_stats()
    print("%-20s spp=%d kernel %8.2f ms | %8.1f Msamples/s | %6.2f Gseg/s | %.2f seg/sample"
          % (name, spp, st.kernel_ms, st.samples / st.kernel_ms / 1e3, st.segments / st.kernel_ms / 1e6,
             st.segments / st.samples), flush=True)
    scene.close()

"""Developer probe: what do the textures of noise_and_textures.yml cost?  Renders the C4
frame with each texture replaced by a SolidColor in turn (pictures differ, timing tells)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd")
host = importlib.import_module("racer-tracer_amd.host")
abi = importlib.import_module("racer-tracer_amd.abi")
s = host.Session(os.path.join(ROOT, "scenes", "config_c4.yml"), scene=os.path.join(ROOT, "scenes", "noise_and_textures.yml"))
p = s.params
p.samples = 128
d = s.desc
names = {0: "solid", 1: "checker", 2: "image", 3: "noise"}
print("textures:", [(i, names[d.textures[i].kind]) for i in range(d.n_textures)])
print("materials:", [(i, d.materials[i].kind, d.materials[i].texture) for i in range(d.n_materials)])


def run(label):
    scene = rt.Scene(s)
    scene.render_frame(s.camera, p)
    scene.render_frame(s.camera, p)
    st = scene.last_stats()
    print("%-28s kernel %7.2f ms  %6.2f Gseg/s  %.2f seg/sample" % (label, st.kernel_ms, st.segments / st.kernel_ms / 1e6, st.segments / st.samples), flush=True)
    scene.close()


run("as shipped")
saved = [d.textures[i].kind for i in range(d.n_textures)]
for kind in (3, 1, 2):
    for i in range(d.n_textures):
        if saved[i] == kind:
            d.textures[i].kind = 0
    run("without %s" % names[kind])
    for i in range(d.n_textures):
        d.textures[i].kind = saved[i]
for i in range(d.n_textures):
    d.textures[i].kind = 0
run("all solid (other variant)")

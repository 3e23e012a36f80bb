"""rt_scene_create + first-frame latency (run on the GPU box).

The reference rebuilds its scene (BVH included) on every object event and the camera changes between calls
(racer-tracer/src/main.rs:174-189, bvh_node.rs:176-205), so a binding pays rt_scene_create + the first render of the
new scene each time.  Per scene: the very first create of the process (cold: code object, context), then five rounds
of  destroy the previous scene -> create -> first rt_render at 1080p (preview scale 4 x 40 spp, what interactive.rs
renders on a change, and a full 1-spp frame) -> second render of the same kind.

    python3 tools/time_scene_create.py [out.txt]
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("racer-tracer_amd")
host = importlib.import_module("racer-tracer_amd.host")
abi = rt.abi


def ms(f):
    t0 = time.perf_counter()
    r = f()
    return (time.perf_counter() - t0) * 1e3, r


def main():
    import ctypes as C
    import numpy as np
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    lib = rt.lib()
    noop = abi.RtTileCallback(lambda *a: None)   # the binding's own work per tile is not the library's latency

    def emit(s=""):
        print(s, flush=True)
        if out:
            out.write(s + "\n")

    cases = [("cornell_box.yml (6 rects)", "cornell_box.yml"), ("cornell_box_boxes.yml (6 rects + 2 wrapped boxes)", "cornell_box_boxes.yml"),
             ("noise_and_textures.yml (4 spheres, 2 MB earth map, Perlin table)", "noise_and_textures.yml"),
             ("random (485 spheres: BVH build)", "random")]
    emit("# rt_scene_create and first-frame latency, 1920x1080, wall-clock ms on the calling thread (min / median of 5 rounds)")
    emit("# 'first' = the first rt_render after the create (allocates slices / pinned frame / counters), 'again' = the next one")
    first_of_process = True
    for title, scene_file in cases:
        path = scene_file if scene_file == "random" else os.path.join(ROOT, "scenes", scene_file)
        session = host.Session(os.path.join(ROOT, "scenes", "config_c2.yml"), scene=path)
        desc = session.desc
        full = abi.RtRenderParams.from_buffer_copy(session.params)
        full.samples = 1
        preview = abi.RtRenderParams.from_buffer_copy(session.params)
        preview.samples, preview.scale = 40, 4          # interactive.rs:196-267's preview renderer
        frame = np.zeros((full.height, full.width, 3), dtype=np.float64)   # caller-owned, touched once: no page faults in the timings
        frame_ptr = frame.ctypes.data_as(C.POINTER(C.c_double))

        def render_tiles(sc, p):
            rt.check(lib.rt_render(sc._h, C.byref(session.camera), C.byref(p), noop, None, None), "rt_render")

        def render_frame(sc, p):
            rt.check(lib.rt_render_frame(sc._h, C.byref(session.camera), C.byref(p), frame_ptr), "rt_render_frame")

        cold, scene = ms(lambda: rt.Scene(desc))
        rows = {"create": [], "destroy": [], "preview first": [], "preview again": [], "frame first": [], "frame again": []}
        for _ in range(5):
            t, _r = ms(scene.close)
            rows["destroy"].append(t)
            t, scene = ms(lambda: rt.Scene(desc))
            rows["create"].append(t)
            t, _r = ms(lambda: render_tiles(scene, preview))
            rows["preview first"].append(t)
            t, _r = ms(lambda: render_tiles(scene, preview))
            rows["preview again"].append(t)
            t, _r = ms(scene.close)
            t, scene = ms(lambda: rt.Scene(desc))
            t, _r = ms(lambda: render_frame(scene, full))
            rows["frame first"].append(t)
            t, _r = ms(lambda: render_frame(scene, full))
            rows["frame again"].append(t)
        scene.close()
        emit()
        emit("%s" % title)
        emit("  first create of %s: %.2f ms" % ("the process (context + code object)" if first_of_process else "this scene", cold))
        for k in ("destroy", "create", "preview first", "preview again", "frame first", "frame again"):
            v = sorted(rows[k])
            emit("  %-14s %8.2f / %8.2f" % (k, v[0], v[len(v) // 2]))
        emit("  create + first preview (what an object event costs): %.2f ms" % (sorted(rows["create"])[2] + sorted(rows["preview first"])[2]))
        first_of_process = False


if __name__ == "__main__":
    main()

"""Developer probe: the several-share calls under stress on ONE card — what a multi-GPU node runs per device, queued behind
each other here.  Per iteration, with 2 or 3 scenes of the same description on device 0 (staged and in-place gather mixed):
rt_render_frame_multi, rt_render_frame_multi_device, rt_render_multi, every frame / tile equal to the single-scene frame;
every third iteration a stream cancelled at a random moment (must return a prefix, must not hang), every fifth a scene is
destroyed and rebuilt (its render buffers come back from the cache).  python3 tools/soak_multi.py [iterations]"""
import importlib, os, random, sys, threading, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes_py as S
rt = importlib.import_module("racer-tracer_amd")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
random.seed(11)
for name, make, w, h, spp in (("cornell_box_boxes", S.cornell_box_boxes, 640, 360, 24), ("three_balls", S.three_balls, 480, 270, 16)):
    bundle, cam, _ = make()
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp, tiles_w=8, tiles_h=5)
    scenes = [rt.Scene(bundle, gather=S.abi.RT_GATHER_STAGED if k % 2 else S.abi.RT_GATHER_AUTO) for k in range(3)]
    want = scenes[0].render_frame(camera, params)
    want_tiles = scenes[0].render_tiles(camera, params)
    out = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    bad = cut = rebuilt = 0
    t0 = time.time()
    for i in range(reps):
        n = 2 + i % 2
        rows = (0, 8, 16, 24)[i % 4]
        bad += not np.array_equal(rt.render_frame_multi(scenes[:n], camera, params, rows), want)
        out.fill_(-1.0)
        torch.cuda.synchronize()
        rt.render_frame_multi_device(scenes[:n], camera, params, out.data_ptr(), rows)
        bad += not np.array_equal(out.cpu().numpy(), want)
        tiles = rt.render_tiles_multi(scenes[:n], camera, params, rows)
        bad += len(tiles) != len(want_tiles) or any(not np.array_equal(a[4], b[4]) for a, b in zip(tiles, want_tiles))
        if i % 3 == 0:
            ev = threading.Event()
            timer = threading.Timer(random.random() * 0.004, ev.set)
            timer.start()
            try:
                part = rt.render_tiles_multi(scenes[:n], camera, params, rows, cancel=ev.is_set)
            except rt.RtError as e:  # the hook was already up when the call began: CancelEvent (cpu.rs:82-85), nothing rendered
                assert e.code == S.abi.RT_ERR_CANCEL_EVENT
                part = []
            timer.join()
            cut += len(part) < len(want_tiles)
            bad += any(not np.array_equal(a[4], b[4]) for a, b in zip(part, want_tiles))
        if i % 5 == 4:
            k = random.randrange(3)
            scenes[k].close()
            scenes[k] = rt.Scene(bundle, gather=S.abi.RT_GATHER_STAGED if k % 2 else S.abi.RT_GATHER_AUTO)
            rebuilt += 1
    print("%-18s %d x (frame_multi + frame_multi_device + render_multi) over 2-3 shares: %d mismatches, %d streams cut short, %d scenes rebuilt, %.1f s"
          % (name, reps, bad, cut, rebuilt, time.time() - t0), flush=True)
    for s in scenes:
        s.close()
    assert bad == 0

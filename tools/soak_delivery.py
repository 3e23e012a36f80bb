"""Developer probe: the delivering launch under stress.  Every frame that rt_render_frame / rt_render hand to the host must
be the two-pass frame bit for bit — the slices cross waves on eight XCDs through sc1 stores / loads and a relaxed counter
(rt_trace_pool_kernel.hip: deliver_item), so a visibility bug would show as a rare stale pixel — with cancels raised at
random moments in between (the per-tile counters of a launch that was cut short must be cleared)."""
import ctypes as C, importlib, os, sys, hashlib, random, threading, time
import numpy as np
import torch
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
rt = importlib.import_module("racer-tracer_amd"); host = importlib.import_module("racer-tracer_amd.host")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
random.seed(3)
for wl, cfg, scene, spp in (("c2", "config_c2.yml", "three_balls.yml", 48), ("c3", "config_c3.yml", "cornell_box.yml", 96),
                            ("c4", "config_c4.yml", "noise_and_textures.yml", 32), ("random", "config_c2.yml", "random", 8),
                            ("boxes", "config_c3.yml", "cornell_box_boxes.yml", 32)):  # (random, boxes: two items in flight per wave)
    path = scene if scene == "random" else os.path.join(ROOT, "scenes", scene)
    s = host.Session(os.path.join(ROOT, "scenes", cfg), scene=path)
    p = s.params
    p.samples = spp
    sc = rt.Scene(s)
    dev = torch.zeros((p.height, p.width, 3), dtype=torch.float64, device="cuda")
    sc.render_frame_device(s.camera, p, dev.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = dev.cpu().numpy()
    bad = cancels = 0
    t0 = time.time()
    for i in range(reps):
        got = sc.render_frame(s.camera, p)
        bad += not np.array_equal(got, want)
        tiles = sc.render_tiles(s.camera, p)
        for r, c, w, h, arr in tiles:
            bad += not np.array_equal(arr, want[r:r + h, c:c + w])
        if i % 3 == 0:  # a cancel somewhere inside the next stream
            ev = threading.Event()
            timer = threading.Timer(random.random() * 0.02, ev.set)
            timer.start()
            part = sc.render_tiles(s.camera, p, cancel=ev.is_set)
            timer.join()
            cancels += len(part) < len(tiles)
            for r, c, w, h, arr in part:
                bad += not np.array_equal(arr, want[r:r + h, c:c + w])
    print("%-7s %d x (rt_render_frame + rt_render [+ a cancelled stream]) at %d spp: %d mismatching frames/tiles, %d streams cut short, %.1f s"
          % (wl, reps, spp, bad, cancels, time.time() - t0), flush=True)
    sc.close()
    assert bad == 0

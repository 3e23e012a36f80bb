"""`python bench.py --gpus N` as the driver runs it: with no launcher around it, bench.py must spawn its N
ranks itself (before touching the GPU), relay ONE JSON line and exit 0.  Rehearsed with N = 2 on one card
(BENCH_REHEARSE_ON_ONE_GPU=1: both ranks on device 0, the strip gather over gloo); the gathered frame must be
bit-identical to a single-rank render.  The child is started in conftest.pytest_sessionstart."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_without_launcher_spawns_ranks_not_an_error():
    """CPU side of the same contract: the parent spawns before importing torch / HIP, so on a box without
    a GPU the failure comes from the RANKS ('needs a GPU'), not from a refusal to launch."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--spp", "1", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    if run.returncode == 0:  # a box with GPUs: a result line must have come back
        assert json.loads(run.stdout.strip().splitlines()[-1])["n_gpus"] == 2
    else:
        assert "must be launched with" not in run.stderr
        assert "needs a GPU" in run.stderr or "device(s) visible" in run.stderr, run.stderr[-2000:]


@pytest.mark.gpu
def test_bench_self_launches_two_ranks(request, gpu):
    proc = getattr(request.config, "_bench_rehearsal", None)
    if proc is None:
        pytest.skip("the rehearsal is started by `pytest -m gpu` (conftest.pytest_sessionstart)")
    try:
        stdout, _ = proc.communicate(timeout=900)
    except subprocess.TimeoutExpired:
        proc.kill()
        raise
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert proc.returncode == 0 and len(lines) == 1, (proc.returncode, stdout[-2000:])
    out = json.loads(lines[0])
    with open(os.path.join(ROOT, "gpurun_out", "bench_rehearsal_gpus2.json"), "w") as f:
        json.dump(out, f, indent=1)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
    assert out["rehearsal_frame_matches_single_rank"] is True
    assert "gloo" in out["config"]["parallelism"] and "RCCL gather" not in out["config"]["parallelism"]   # say what ran
    assert out["config"]["workload"].startswith("cornell_box.yml 1920x1080 64spp")


@pytest.mark.gpu
def test_rccl_process_group_and_f64_gather_with_one_rank(request, gpu):
    """`bench.py --force-dist`: backend "nccl" (= RCCL) initialised on the real device, the strips of the frame sent
    through dist.gather as f64 device tensors, the timings all-reduced; the gathered frame equals a plain render."""
    proc = getattr(request.config, "_bench_force_dist", None)
    if proc is None:
        pytest.skip("started by `pytest -m gpu` (conftest.pytest_sessionstart)")
    try:
        stdout, _ = proc.communicate(timeout=900)
    except subprocess.TimeoutExpired:
        proc.kill()
        raise
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert proc.returncode == 0 and len(lines) == 1, (proc.returncode, stdout[-2000:],
                                                      open(os.path.join(ROOT, "gpurun_out", "bench_force_dist.stderr.log")).read()[-3000:])
    out = json.loads(lines[0])
    with open(os.path.join(ROOT, "gpurun_out", "bench_force_dist.json"), "w") as f:
        json.dump(out, f, indent=1)
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert out["force_dist"].startswith("backend nccl, world 1")
    assert "RCCL gather" in out["config"]["parallelism"]
    assert out["rehearsal_frame_matches_single_rank"] is True


@pytest.mark.gpu
def test_cli_renders_and_saves_the_reference_png(request, gpu, orc, rt):
    """racer-tracer-amd -c config -s scene --image-action png: render, tone map, `<SHA-256>.png` (main.rs:148-158,
    png.rs:19-55); the picture is the oracle's.  With --devices 2 the frame is sharded over two GPUs of the process
    (rt_render_frame_multi) - or, on a one-GPU box, refused with the device-index error, not a crash."""
    import importlib
    import numpy as np
    from PIL import Image
    if request.config._cli_run is None:
        pytest.skip("the CLI is started by `pytest -m gpu` (conftest.pytest_sessionstart)")
    out, run = request.config._cli_run
    assert run.returncode == 0 and "Saved image to" in run.stderr, run.stderr[-2000:]
    pngs = sorted(f for f in os.listdir(out) if f.endswith(".png") and len(f) == 68)
    host = importlib.import_module("racer-tracer_amd.host")
    s = host.Session(os.path.join(ROOT, "scenes", "config_c1.yml"), scene=os.path.join(ROOT, "scenes", "three_balls.yml"))
    ref, _ = orc.render(s.desc, s.camera, s.params)
    want = orc.pack_rgba8(s.tone_map(ref))
    two = request.config._cli_two_devices
    if rt.device_count() >= 2:
        assert two.returncode == 0 and len(pngs) == 1     # the same bytes -> the same SHA-256 name
    else:
        assert two.returncode == rt.abi.RT_ERR_INVALID_ARGUMENT and "device index" in two.stderr and len(pngs) == 1
    img = np.array(Image.open(os.path.join(out, pngs[0])))
    assert img.shape == want.shape == (225, 400, 4)
    assert np.abs(img.astype(int) - want.astype(int)).max() <= 1 and (img != want).mean() < 1e-3
    assert pngs[0] == host.sha256_hex(img.tobytes()) + ".png"

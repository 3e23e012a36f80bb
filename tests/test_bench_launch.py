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
    assert out["config"]["workload"].startswith("cornell_box.yml 1920x1080 64spp")

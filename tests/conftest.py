import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """CPU oracle (test infrastructure), built on demand with gcc."""
    from oracle import oracle_ctypes
    oracle_ctypes.lib()
    return oracle_ctypes


@pytest.fixture(scope="session")
def rt():
    """The shipped ctypes binding; the library must already be built."""
    return importlib.import_module("racer-tracer_amd")


@pytest.fixture(scope="session")
def host():
    """The ctypes binding of the C++ host layer (include/rt_host.h)."""
    return importlib.import_module("racer-tracer_amd.host")


@pytest.fixture(scope="session")
def abi():
    return importlib.import_module("racer-tracer_amd.abi")


@pytest.fixture(scope="session")
def gpu(rt):
    """Skips nothing: a -m gpu run without a device is a failure, not a skip."""
    n = rt.device_count()
    assert n > 0, "pytest -m gpu needs a visible HIP device (rt_device_count() == 0)"
    return n


# ---- bench.py --gpus 2, self-launched, on ONE card ---------------------------------------------
# A process that has initialised the GPU must not start other GPU programs carelessly on this pool, and
# pytest's process initialises it with the first `gpu` fixture.  So the rehearsal of `python bench.py
# --gpus 2` (the command the driver runs; bench.py spawns its two ranks itself) is started HERE, before
# any test runs, and tests/test_bench_launch.py only collects its result.
def pytest_sessionstart(session):
    session.config._bench_rehearsal = None
    session.config._bench_force_dist = None
    session.config._cli_run = None
    if (session.config.getoption("-m") or "").strip() != "gpu":
        return
    import subprocess
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    log = open(os.path.join(out_dir, "bench_rehearsal_gpus2.stderr.log"), "w")
    env = dict(os.environ, BENCH_REHEARSE_ON_ONE_GPU="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--spp", "64", "--no-cpu-baseline"]
    session.config._bench_rehearsal = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=log, env=env, text=True, cwd=ROOT)
    # ... and RCCL for real: `bench.py --force-dist` = torch.distributed.run with ONE rank, init_process_group("nccl"),
    # the strip gather forced through dist.gather on the f64 device tensors, an all_reduce of the timings.  What a
    # multi-GPU node would otherwise be the first to find out: librccl loads, device_id= is accepted, f64 gathers work.
    log2 = open(os.path.join(out_dir, "bench_force_dist.stderr.log"), "w")
    env2 = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_REHEARSE_ON_ONE_GPU"):
        env2.pop(k, None)
    cmd2 = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "2", "--warmup", "1",
            "--spp", "64", "--no-cpu-baseline", "--pmc", "none"]
    session.config._bench_force_dist = subprocess.Popen(cmd2, stdout=subprocess.PIPE, stderr=log2, env=env2, text=True, cwd=ROOT)
    # ... and the headless CLI end to end (-c/-s/--image-action png, the reference's flags), for the same reason
    import tempfile
    out = tempfile.mkdtemp(prefix="rt_cli_")
    exe = os.path.join(ROOT, "racer-tracer_amd", "bin", "racer-tracer-amd")
    cli = [exe, "-c", os.path.join(ROOT, "scenes", "config_c1.yml"), "-s", os.path.join(ROOT, "scenes", "three_balls.yml"),
           "--image-action", "png", "--seed", "1"]
    session.config._cli_run = (out, subprocess.run(cli, capture_output=True, text=True, cwd=out, timeout=600))
    session.config._cli_two_devices = subprocess.run(cli + ["--devices", "2"], capture_output=True, text=True, cwd=out, timeout=600)


def pytest_sessionfinish(session, exitstatus):
    for name in ("_bench_rehearsal", "_bench_force_dist"):
        proc = getattr(session.config, name, None)
        if proc is not None and proc.poll() is None:  # the collecting test did not run (e.g. -x stopped earlier)
            proc.kill()
            proc.wait()

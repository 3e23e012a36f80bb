#!/usr/bin/env python3
"""bench.py — Msamples/s of the MI355X render path on BASELINE.json's workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4]

A "step" is one full render of the workload frame: rt_render_frame_device on
this rank's row strips (+ the RCCL gather of finished strips to rank 0 when
N > 1) and the tone-map + RGBA8 pass over the finished frame (SURVEY 8(d):
"resolve/tone-map kernel included").  Scene, camera and output buffers are
resident in HBM before the timed region starts.  Rank 0 prints ONE JSON line.

Default workload = BASELINE.json configs[2] (cornell_box.yml, 1920x1080, 1024
spp, max_depth 20, Aces): it is the configuration the north star's target is
quoted on and it fits one GPU.  `--workload c2|c4|c5` select configs[1]/[3]/[4].

Scaling is STRONG: the frame is fixed, N GPUs split its rows (8-row strips,
interleaved), so value = W*H*spp / time-of-the-slowest-rank.

`--gpus N` with N > 1 works both ways: under torch.distributed.run (RANK /
WORLD_SIZE in the environment) this process is one rank; started plainly
(`python bench.py --gpus N`) it spawns the N ranks as child processes BEFORE
touching the GPU, relays rank 0's JSON line and exits with their status.

roofline: the trace kernel keeps ray state in registers, so HBM is not what
bounds it; the bound is the vector ALU.  `frac` = counted f64 FLOP per second /
the 78.6 TFLOP/s f64 vector peak, `issue_frac` = sum over instruction classes of
count x the class's minimum issue cost / the SIMD cycles of the launch — two
numbers that cannot exceed 1.  The counters are collected LIVE: before this
process touches the GPU it runs rocprofv3 --pmc passes over itself (tools/pmc.py,
one pass per counter group, about ten seconds each); `--pmc file` uses the
committed summary under profiles/ instead (only while its source stamp matches
the tree), `--pmc none` skips counters.  SURVEY 8(d)'s algorithmic-bytes figure
is kept as `hbm_algorithmic`.

Also in the line: `host_delivered` — W*H*spp over the wall time of rt_render
with the reference's 10x10 tile grid (every tile copied out of the callback),
and of rt_render_frame into a caller-owned host frame: what a binding of the
reference receives, next to the device-resident `value`.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

sys.path.insert(0, os.path.join(ROOT, "tools"))
from source_stamp import kernel_source_sha, library_sha  # noqa: E402

STRIP_ROWS = 8
PMC_ROUND = "r04"              # profiles/<round>_<workload>_pmc_summary.json
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
SIMDS = 256 * 4                # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4                # MI355X_MICROARCH.md peak engine clock
F64_VECTOR_PEAK_TFLOPS = 78.6  # half the 157.3 TFLOP/s f32 vector peak (MI355X_MICROARCH.md): 1024 SIMDs x 16 lanes x 2 x 2.4 GHz
BYTES_PER_SEGMENT_F64 = 192.0  # 96-B f64 ray record read + written per segment
# Minimum SIMD cycles one wave instruction of a class holds its SIMD's vector issue (CDNA4: a SIMD executes 32 lanes
# of 32-bit or 16 lanes of 64-bit work per cycle; measured per instruction in profiles/r02_valu_cost.txt: f64
# add/mul/fma 4.2, v_rcp/rsq/sqrt_f64 16.1, plain 32-bit VOP2 2.3, 32-bit multiplies and v_mad_u64_u32 4.2-4.3,
# conversions 4.1-4.2).  The floor of each class is used, so the sum cannot exceed the cycles that exist.
ISSUE_COST = {"f64": 4.0, "trans_f64": 16.0, "f32": 2.0, "trans_f32": 8.0, "int32": 2.0, "int64": 4.0, "cvt": 4.0, "other": 2.0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c5", "random", "boxes", "emissive"])
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (debug)")
    ap.add_argument("--seed", type=int, default=0, help="override the RNG seed of the config (SURVEY 8(d): seeds 2, 3 for variance)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--pmc", default="live", choices=["live", "file", "none"],
                    help="where the roofline's counters come from: rocprofv3 passes over this command run now (live), "
                         "the committed profiles/ summary (file), or nowhere (none)")
    ap.add_argument("--pmc-child", action="store_true", help="internal: one short run under rocprofv3 (tools/pmc.py)")
    ap.add_argument("--no-host-delivery", action="store_true", help="skip the rt_render / rt_render_frame host timings")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the strip gather as a collective even with "
                         "ONE rank (proves librccl + an f64 device gather on a one-GPU box)")
    args = ap.parse_args()
    if args.pmc_child:
        args.steps, args.warmup, args.no_cpu_baseline, args.pmc, args.no_host_delivery = 1, 1, True, "none", True
    return args


def load_workload(host, name, spp_override):
    table = {
        "c2": ("three_balls.yml", "config_c2.yml"),
        "c3": ("cornell_box.yml", "config_c3.yml"),
        "c4": ("noise_and_textures.yml", "config_c4.yml"),
        "c5": ("cornell_box.yml", "config_c5.yml"),   # configs[4]: 3840x2160x4096, meant for 8 GPUs
        # not BASELINE configs: the other kernel variants at 1080p, for counters and profiles
        "random": ("random", "config_c2.yml", 64),                 # 485 spheres, BVH variant
        "boxes": ("cornell_box_boxes.yml", "config_c3.yml", 128),  # PRIMS_ANY linear loop
        "emissive": ("emissive.yml", "config_c2.yml", 128),        # PRIMS_ANY + Noise
    }
    scene_file, config_file = table[name][:2]
    scene_arg = scene_file if scene_file == "random" else os.path.join(ROOT, "scenes", scene_file)
    session = host.Session(os.path.join(ROOT, "scenes", config_file), scene=scene_arg)
    if len(table[name]) > 2:
        session.params.samples = table[name][2]
    if spp_override:
        session.params.samples = spp_override
    return session, "%s %dx%d %dspp max_depth %d" % (
        scene_file, session.params.width, session.params.height, session.params.samples,
        session.params.max_depth)


def usable_cores():
    """CPUs this process may really use: online cores capped by the cgroup quota."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(session, seconds):
    """Times the CPU oracle (port of CpuRenderer: recursive f64, 10x10 tiles on a
    thread pool) on a bounded sample of the same workload: same scene and
    resolution, reduced spp, one thread per usable core."""
    from oracle import oracle_ctypes as orc
    abi = importlib.import_module("racer-tracer_amd.abi")
    cores = min(usable_cores(), max(1, session.params.tiles_w) * max(1, session.params.tiles_h))
    p = abi.RtRenderParams.from_buffer_copy(session.params)
    p.strip_count = 0
    p.samples = 4
    t0 = time.time()
    orc.render(session.desc, session.camera, p, n_threads=cores)
    t1 = max((time.time() - t0) / 4.0, 1e-3)
    spp = int(max(1, min(512, seconds / t1)))
    p.samples = spp
    t0 = time.time()
    _, segs = orc.render(session.desc, session.camera, p, n_threads=cores)
    dt = time.time() - t0
    n = p.width * p.height * spp
    return {"value": round(n / dt / 1e6, 3), "unit": "Msamples/s",
            "cores": cores,
            "online_cores": orc.lib().orc_online_cores(), "kind": "port",
            "sample": "same scene and %dx%d frame at %d spp (%.1f s, %.2f segments/sample)"
                      % (p.width, p.height, spp, dt, segs / n)}


def trace_kernel_counters(summary, workload_name, origin):
    """The trace kernel's counters out of a tools/pmc.py summary, or (None, reason)."""
    stamp = summary.get("_stamp", {})
    if stamp.get("source_sha") != kernel_source_sha():
        return None, "%s was collected on other kernel sources (stamp mismatch): re-run tools/pmc.py" % origin
    if stamp.get("library_sha") is not None and stamp.get("library_sha") != library_sha():
        # a committed summary travels to another box, where the library is rebuilt: only a LIVE summary is held to the file
        if origin.startswith("live"):
            return None, "%s was collected on another build of the library" % origin
    if stamp.get("workload") != workload_name:
        return None, "%s is for %r" % (origin, stamp.get("workload"))
    kernels = [k for k in summary if "k_trace_pool_f64" in k]
    if len(kernels) != 1:
        return None, "%s holds %d trace kernels" % (origin, len(kernels))
    c = dict(summary[kernels[0]], _origin=origin, _stamp=stamp, _kernel=kernels[0])
    for need in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"):
        if need not in c:
            return None, "%s lacks %s (%s)" % (origin, need, "; ".join(stamp.get("errors", [])) or "pass missing")
    return c, None


def committed_pmc_summary(workload, workload_name):
    """profiles/<round>_<workload>_pmc_summary.json (tools/pmc.py), refused when it was collected on other kernel
    sources than the ones in the tree or on another frame."""
    path = os.path.join(ROOT, "profiles", "%s_%s_pmc_summary.json" % (PMC_ROUND, workload))
    rel = os.path.relpath(path, ROOT)
    if not os.path.exists(path):
        return None, "no %s" % rel
    with open(path) as f:
        return trace_kernel_counters(json.load(f), workload_name, rel)


def live_pmc_summary(args):
    """rocprofv3 --pmc passes over `bench.py --pmc-child` with this run's workload, as child processes of a parent
    that has not touched the GPU yet (tools/pmc.py).  Returns the summary dict or (None, reason)."""
    import pmc
    if args.gpus != 1 or "WORLD_SIZE" in os.environ:
        return None, "PMC passes are single-GPU"
    import shutil
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 is not on PATH"
    child = ["--workload", args.workload] + (["--spp", str(args.spp)] if args.spp else []) + (["--seed", str(args.seed)] if args.seed else [])
    t0 = time.time()
    summary = pmc.collect(child, log=lambda m: sys.stderr.write("bench.py: %s\n" % m), budget_s=150, pass_timeout=60)
    summary["_stamp"]["collected_in_s"] = round(time.time() - t0, 1)
    return summary, None


def mean_of(c, name):
    return c[name]["mean"] if name in c else None


def valu_roofline(c, kernel_ms, segments):
    """Vector-ALU view of one launch from the SQ counters (per-dispatch means) and the LIVE kernel duration.
    frac       = f64 FLOP/s over the f64 vector peak: (ADD + MUL + TRANS + 2 FMA) f64 wave-instructions x the mean
                 active lanes per vector instruction (the counters give no per-class lane count) / kernel time.
    issue_frac = sum over classes of instructions x the class's MINIMUM issue cost (ISSUE_COST) over the SIMD-cycles
                 of the launch at the peak clock: the share of the launch in which the vector pipes were provably
                 occupied.  Both are <= 1 by construction.
    valu_busy  = 4 x SQ_ACTIVE_INST_VALU / SIMD-cycles: the hardware's own busy figure; it books a quad-cycle per
                 instruction although 32-bit ones issue in 2, so it can pass 1 (round 2 printed it as `frac`).
    scalar_busy = SQ_ACTIVE_INST_SCA / CU-cycles: a CU has ONE scalar ALU for its four SIMDs; C3 kept it 67 % busy beside
                 vector pipes at 98 % in round 3, and taking scalar instructions out is what moved the frame in round 4."""
    secs = kernel_ms * 1e-3
    simd_cycles = SIMDS * CLOCK_GHZ * 1e9 * secs
    total = c["SQ_INSTS_VALU"]["mean"]
    lanes = c["SQ_THREAD_CYCLES_VALU"]["mean"] / c["SQ_ACTIVE_INST_VALU"]["mean"]
    out = {"valu_busy": round(4.0 * c["SQ_ACTIVE_INST_VALU"]["mean"] / simd_cycles, 4),
           "lanes_per_inst": round(lanes, 1),
           "valu_insts_per_launch": total,
           "valu_insts_per_segment": round(total / segments, 3)}
    if mean_of(c, "SQ_ACTIVE_INST_SCA") is not None:
        out["scalar_busy"] = round(c["SQ_ACTIVE_INST_SCA"]["mean"] / (simd_cycles / 4.0), 4)
        if mean_of(c, "SQ_INSTS_SALU") is not None:
            out["salu_insts_per_segment"] = round(c["SQ_INSTS_SALU"]["mean"] / segments, 3)
    if mean_of(c, "SQ_WAVE_CYCLES") and mean_of(c, "SQ_WAIT_INST_ANY") is not None:
        out["wave_time_waiting_frac"] = round(c["SQ_WAIT_INST_ANY"]["mean"] / c["SQ_WAVE_CYCLES"]["mean"], 4)
        if mean_of(c, "SQ_ACTIVE_INST_ANY") is not None:
            out["wave_time_issuing_frac"] = round(c["SQ_ACTIVE_INST_ANY"]["mean"] / c["SQ_WAVE_CYCLES"]["mean"], 4)
    f64 = [mean_of(c, "SQ_INSTS_VALU_%s_F64" % k) for k in ("ADD", "MUL", "FMA", "TRANS")]
    if None in f64:
        out["frac"] = out["achieved"] = None
        out["per_class_counters"] = "missing: %s" % "; ".join(c["_stamp"].get("errors", []))
        return out
    add, mul, fma, trans = f64
    flops = (add + mul + trans + 2.0 * fma) * lanes
    achieved = flops / secs / 1e12
    classes = {"f64": add + mul + fma, "trans_f64": trans}
    f32 = [mean_of(c, "SQ_INSTS_VALU_%s_F32" % k) or 0.0 for k in ("ADD", "MUL", "FMA")]
    classes["f32"] = sum(f32)
    classes["trans_f32"] = mean_of(c, "SQ_INSTS_VALU_TRANS_F32") or 0.0
    classes["int32"] = mean_of(c, "SQ_INSTS_VALU_INT32") or 0.0
    classes["int64"] = mean_of(c, "SQ_INSTS_VALU_INT64") or 0.0
    classes["cvt"] = mean_of(c, "SQ_INSTS_VALU_CVT") or 0.0
    classes["other"] = max(0.0, total - sum(classes.values()))  # moves, selects, compares, lane ops
    issue_cycles = sum(n * ISSUE_COST[k] for k, n in classes.items())
    out.update({"achieved": round(achieved, 3), "peak": F64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / F64_VECTOR_PEAK_TFLOPS, 4),
                "issue_frac": round(issue_cycles / simd_cycles, 4),
                # ... of which the share spent on lanes that were enabled (the rest ran masked off: divergence)
                "useful_lane_issue_frac": round(issue_cycles / simd_cycles * lanes / 64.0, 4),
                "f64_flop_per_segment": round(flops / segments, 2),
                "valu_insts_by_class": {k: round(v / total, 4) for k, v in classes.items()},
                "issue_cost_cycles": ISSUE_COST})
    return out


def hbm_traffic(c):
    """HBM bytes of one launch from the FETCH_SIZE / WRITE_SIZE passes, corrected as
    MI355X_MICROARCH.md prescribes: units of 1 KiB, reads x2 on gfx950 (128-B requests counted as 64 B)."""
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    return 2.0 * c["FETCH_SIZE"]["mean"] * 1024.0 + c["WRITE_SIZE"]["mean"] * 1024.0


def host_delivery(rt, scene, session, reps=5):
    """What a binding of the reference receives: the wall time of rt_render (cpu.rs:73-115's tile grid, every tile
    copied out of the callback into a host frame, as `rgb.to_vec()` would) and of rt_render_frame into a caller-owned
    frame.  Pinned staging is the library's business; the destination here is ordinary pageable memory."""
    import ctypes as C
    import numpy as np
    abi = rt.abi
    p = abi.RtRenderParams.from_buffer_copy(session.params)
    p.strip_count = 0
    W, H = p.width, p.height
    frame = np.zeros((H, W, 3), dtype=np.float64)
    # one buffer per tile, filled by a plain memcpy inside the callback — what `rgb.to_vec()` costs a Rust binding
    # (INTEGRATION.md section 3); the reference's own tiles are moved, not copied (cpu.rs:64-70)
    tile_bytes = (W // p.tiles_w + W % p.tiles_w) * (H // p.tiles_h + H % p.tiles_h) * 24
    store = np.zeros((p.tiles_w * p.tiles_h, tile_bytes), dtype=np.uint8)
    slots = [store[i].ctypes.data for i in range(store.shape[0])]
    tiles = []

    def on_tile(_user, rgb, r, c, w, h):
        C.memmove(slots[len(tiles)], rgb, w * h * 24)
        tiles.append((r, c, w, h))

    cb = abi.RtTileCallback(on_tile)
    lib = rt.lib()
    out = {}
    for name, call in (("rt_render", lambda: lib.rt_render(scene._h, C.byref(session.camera), C.byref(p), cb, None, None)),
                       ("rt_render_frame", lambda: lib.rt_render_frame(scene._h, C.byref(session.camera), C.byref(p),
                                                                     frame.ctypes.data_as(C.POINTER(C.c_double))))):
        rt.check(call(), name)  # warm-up (pinned frame, counters)
        times = []
        for _ in range(reps):
            del tiles[:]
            t0 = time.perf_counter()
            rt.check(call(), name)
            times.append(time.perf_counter() - t0)
        ms = sum(times) / len(times) * 1e3
        out[name] = {"ms": round(ms, 3), "value": round(W * H * p.samples / (ms * 1e-3) / 1e6, 2), "unit": "Msamples/s",
                     "calls": reps}
        if name == "rt_render":
            out[name]["tiles"] = "%dx%d grid, %d callbacks per call" % (p.tiles_w, p.tiles_h, len(tiles))
    return out


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (nothing
    in this process has touched the GPU), relay rank 0's JSON line, return the children's status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC; with the legacy mode RCCL's
    # (and torch's) cross-process sharing of device memory fails with `hipIpcGetMemHandle: invalid argument`.
    # The image exports it already; it is repeated here so the ranks have it under any launcher environment.
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith("{") and line is None:
            line = out
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited without a result line\n")
        rc = 1
    return rc


def main():
    t_start = time.time()
    args = parse()
    if (args.gpus > 1 or args.force_dist) and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))  # before torch / HIP are even imported
    # LIVE counters: rocprofv3 passes over this command, as children, before this process touches the GPU
    live_summary, live_why_not = (None, "--pmc %s" % args.pmc)
    if args.pmc == "live":
        live_summary, live_why_not = live_pmc_summary(args)
    t_pmc = time.time() - t_start
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: there is no CPU fallback for the render path")
    if world > torch.cuda.device_count() and os.environ.get("BENCH_REHEARSE_ON_ONE_GPU") != "1":
        sys.exit("bench.py --gpus %d: only %d device(s) visible (BENCH_REHEARSE_ON_ONE_GPU=1 runs a "
                 "functional rehearsal of the N > 1 path on one card)" % (world, torch.cuda.device_count()))
    # BENCH_REHEARSE_ON_ONE_GPU=1: every rank uses device 0 and the collective goes
    # over gloo (RCCL cannot put several ranks on one card).  A functional rehearsal of
    # the N > 1 path for a 1-GPU box; its numbers mean nothing.
    rehearsal = os.environ.get("BENCH_REHEARSE_ON_ONE_GPU") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    rt = importlib.import_module("racer-tracer_amd")
    host = importlib.import_module("racer-tracer_amd.host")
    session, workload = load_workload(host, args.workload, args.spp)
    if args.seed:
        session.params.seed = args.seed
    p = session.params
    W, H, spp = p.width, p.height, p.samples
    p.strip_rows, p.strip_count, p.strip_index = STRIP_ROWS, world, rank

    scene = rt.Scene(session, device=local)
    strips = importlib.import_module("racer-tracer_amd.strips")
    gatherer = strips.StripGather(H, W, STRIP_ROWS, world, rank, "cuda", dist, always_collective=args.force_dist)
    # when a collective runs the frame lives in the gather's staging buffer: no copies besides the strips
    frame = gatherer.frame() if gatherer.collective else torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream()
    rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") if rank == 0 else None
    kernel_ms, segments = [], []

    def step(record):
        # trace + resolve on torch's current stream, then the one collective of the path, then (on the rank that
        # holds the whole frame) ScreenBuffer::update's tone map + SavePng's RGBA8 packing in one pass
        scene.render_frame_device(session.camera, p, frame.data_ptr(), stream.cuda_stream)
        gatherer.gather(frame)
        if rank == 0:
            scene.post_rgba8_device(session.tone_map_desc, frame.data_ptr(), W * H, rgba.data_ptr(), None, stream.cuda_stream)
        if record:
            st = scene.last_stats()  # HIP events on the launch stream
            kernel_ms.append(st.kernel_ms)
            segments.append(int(st.segments))

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        # reading the HIP events of a step waits for it: every step at N = 1 (the roofline's kernel time is
        # the average over the timed region), only the last one at N > 1, where the host must run ahead
        step(world == 1 or k == args.steps - 1)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        red_dev = "cpu" if rehearsal else "cuda"
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        seg_t = torch.tensor([float(sum(segments))], dtype=torch.float64, device=red_dev)
        dist.all_reduce(seg_t, op=dist.ReduceOp.SUM)
        total_segments = float(seg_t.item())
        k_t = torch.tensor([sum(kernel_ms)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(k_t, op=dist.ReduceOp.MAX)
        kernel_total_ms = float(k_t.item())
    else:
        total_segments = float(sum(segments))
        kernel_total_ms = sum(kernel_ms)
    recorded = len(kernel_ms)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = W * H * spp / (elapsed / args.steps) / 1e6
        seg_per_step = total_segments / recorded
        k_ms = kernel_total_ms / recorded
        algorithmic = BYTES_PER_SEGMENT_F64 * seg_per_step / (k_ms * 1e-3) / 1e9 / world  # GB/s per GPU
        counters, why_not, origin = None, live_why_not, None
        if live_summary is not None:
            counters, why_not = trace_kernel_counters(live_summary, workload, "live rocprofv3 passes of this run")
            origin = "live"
        if counters is None and args.pmc != "none" and world == 1:
            why_live = why_not
            counters, why_not = committed_pmc_summary(args.workload, workload)
            origin = "file"
            if counters is None:
                why_not = "live: %s; file: %s" % (why_live, why_not)
        out = {
            "metric": "Msamples/s (W*H*spp/s) at %dx%d" % (W, H),
            "value": round(value, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (the reference's own scene YAML; counter-based RNG, seed %d)" % p.seed,
            "config": {"workload": workload, "strip_rows": STRIP_ROWS,
                       "parallelism": "image rows interleaved over %d GPU(s)%s"
                                      % (world, "" if dist is None else
                                         (", gloo gather to rank 0 (one-card rehearsal, NOT RCCL)" if rehearsal
                                          else ", RCCL gather to rank 0"))},
            "step": "rt_render_frame_device (trace + resolve)%s + rt_post_rgba8_device (tone map %s + RGBA8)"
                    % (" + gather" if dist is not None else "", session.tone_map_name),
            "roofline": {
                "bound": "valu",
                "achieved": None, "peak": F64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None,
                "traffic": None,
                "kernel": "k_trace_pool_f64",
                "kernel_ms": round(k_ms, 3),
                "segments_per_launch": seg_per_step / world,
                "gsegments_per_s": round(seg_per_step / (k_ms * 1e-3) / 1e9, 3),
                "hbm_algorithmic": {
                    "bytes_per_segment": BYTES_PER_SEGMENT_F64,
                    "bytes_per_launch": BYTES_PER_SEGMENT_F64 * seg_per_step / world,
                    "achieved": round(algorithmic, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(algorithmic / HBM_PEAK_GBS, 4),
                    "note": "SURVEY 8(d): a 96-B f64 ray record read + written per segment, over the kernel "
                            "time.  NOT a bound of this kernel: ray state lives in registers and is never "
                            "streamed, so the figure can exceed 1; it says how the segment rate compares with "
                            "the best an HBM-streaming wavefront tracer could reach (41.7 G segments/s)"},
            },
        }
        if counters is not None:
            out["roofline"].update(valu_roofline(counters, k_ms, seg_per_step))
            out["roofline"]["counters"] = {
                "origin": origin, "source": counters["_origin"], "kernel": counters["_kernel"],
                "source_stamp": counters["_stamp"]["source_sha"][:12],
                "kernel_ms_under_pmc": counters["_stamp"].get("kernel_ms_under_pmc"),
                "collected_in_s": counters["_stamp"].get("collected_in_s"),
                "how": "rocprofv3 --pmc, one pass per counter group (tools/pmc.py), per-dispatch means"}
            traffic = hbm_traffic(counters)
            if traffic is not None:
                # measured HBM bytes per launch over the live kernel duration
                out["roofline"]["traffic"] = round(traffic / (k_ms * 1e-3) / 1e9, 3)
                out["roofline"]["traffic_bytes_per_launch"] = traffic
                out["roofline"]["traffic_frac_of_hbm_peak"] = round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        else:
            out["roofline"]["counters_unavailable"] = why_not
        t_host = time.time()
        if world == 1 and dist is None and not args.no_host_delivery:
            out["host_delivered"] = host_delivery(rt, scene, session)
            # the reference's own boundary (10x10 tile stream into host memory) as one number next to `value`
            out["value_host_delivered"] = out["host_delivered"]["rt_render"]["value"]
            out["host_delivered"]["vs_device_resident"] = {
                k: round(out["host_delivered"][k]["ms"] / ms_per_step, 4) for k in ("rt_render", "rt_render_frame")}
        t_host = time.time() - t_host
        t_cpu = time.time()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(session, args.cpu_seconds)
        # where the wall time of this command went (the timed region is `steps` x `ms_per_step`; the rest is evidence)
        out["wall_s"] = {"live_counter_passes": round(t_pmc, 1), "timed_region": round(elapsed, 2),
                         "host_delivery_calls": round(t_host, 1), "cpu_baseline": round(time.time() - t_cpu, 1),
                         "total": round(time.time() - t_start, 1)}
        if rehearsal:
            out["rehearsal"] = "N ranks on ONE GPU over gloo: functional check only"
        if args.force_dist:
            out["force_dist"] = "backend %s, world %d: process group initialised, strips gathered with dist.gather" % (
                dist.get_backend(), world)
        if dist is not None:
            # whenever a collective assembled the frame: it must equal a single-rank render of the same frame, bit for bit
            # (outside the timed region; on a real multi-GPU node this is the first evidence that N ranks + RCCL deliver
            # the frame one GPU renders — the strips are keyed by global pixel index, section 5 of DESIGN.md)
            p.strip_count = 0
            whole = torch.zeros_like(frame)
            scene.render_frame_device(session.camera, p, whole.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            out["gathered_frame_matches_single_rank"] = bool(torch.equal(whole, frame))
            if rehearsal or args.force_dist:
                out["rehearsal_frame_matches_single_rank"] = out["gathered_frame_matches_single_rank"]
        print(json.dumps(out), flush=True)
    scene.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

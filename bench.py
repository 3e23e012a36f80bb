#!/usr/bin/env python3
"""bench.py — Msamples/s of the MI355X render path on BASELINE.json's workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4]

A "step" is one full render of the workload frame: rt_render_frame_device on
this rank's row strips (+ the RCCL gather of finished strips to rank 0 when
N > 1).  Scene, camera and output buffer are resident in HBM before the timed
region starts.  Rank 0 prints ONE JSON line.

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
bounds it; the bound reported is VALU issue (SIMD cycles with a vector
instruction executing, from rocprofv3 SQ counters of the same workload under
profiles/, used only while their source stamp matches the kernel sources in the
tree).  SURVEY 8(d)'s algorithmic-bytes figure is kept as `hbm_algorithmic`.
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
from source_stamp import kernel_source_sha  # noqa: E402

STRIP_ROWS = 8
PMC_ROUND = "r02"              # profiles/<round>_<workload>_pmc_summary.json
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
SIMDS = 256 * 4                # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4                # MI355X_MICROARCH.md peak engine clock
BYTES_PER_SEGMENT_F64 = 192.0  # 96-B f64 ray record read + written per segment


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the strip gather as a collective even with "
                         "ONE rank (proves librccl + an f64 device gather on a one-GPU box)")
    return ap.parse_args()


def load_workload(host, name, spp_override):
    table = {
        "c2": ("three_balls.yml", "config_c2.yml"),
        "c3": ("cornell_box.yml", "config_c3.yml"),
        "c4": ("noise_and_textures.yml", "config_c4.yml"),
        "c5": ("cornell_box.yml", "config_c5.yml"),   # configs[4]: 3840x2160x4096, meant for 8 GPUs
    }
    scene_file, config_file = table[name]
    session = host.Session(os.path.join(ROOT, "scenes", config_file),
                           scene=os.path.join(ROOT, "scenes", scene_file))
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


def pmc_summary(workload, workload_name, world):
    """The committed rocprofv3 PMC summary of this workload's trace kernel (tools/gpu_pmc.sh ->
    profiles/<round>_<workload>_pmc_summary.json), or (None, reason).  Refused when it was collected
    on other kernel sources than the ones in the tree, on another frame, or for N > 1."""
    path = os.path.join(ROOT, "profiles", "%s_%s_pmc_summary.json" % (PMC_ROUND, workload))
    rel = os.path.relpath(path, ROOT)
    if world != 1:
        return None, "PMC passes are single-GPU"
    if not os.path.exists(path):
        return None, "no %s" % rel
    with open(path) as f:
        summary = json.load(f)
    stamp = summary.get("_stamp", {})
    if stamp.get("source_sha") != kernel_source_sha():
        return None, "%s was collected on other kernel sources (stamp mismatch): re-run tools/gpu_pmc.sh" % rel
    if stamp.get("workload") != workload_name:
        return None, "%s is for %r" % (rel, stamp.get("workload"))
    kernels = [k for k in summary if "k_trace_pool_f64" in k]
    if len(kernels) != 1:
        return None, "%s holds %d trace kernels" % (rel, len(kernels))
    return dict(summary[kernels[0]], _path=rel, _stamp=stamp), None


def valu_roofline(c, kernel_ms, segments):
    """VALU-issue view of one launch.  SQ_ACTIVE_INST_VALU counts, per SIMD, cycles/4 with a vector
    instruction executing (MI355X_MICROARCH.md, SQ counters): x4 = busy SIMD-cycles.  Peak = every SIMD
    busy every cycle of the LIVE kernel duration at the peak clock.  (Four cycles are booked per instruction,
    but plain 32-bit VOP2 instructions issue in 2.3 — tools/microbench/valu_cost.hip — so two waves' bookings
    can overlap and the fraction can pass 1 by a few per cent at six waves per SIMD: it means "at the limit".)"""
    busy_cycles = 4.0 * c["SQ_ACTIVE_INST_VALU"]["mean"]
    secs = kernel_ms * 1e-3
    achieved = busy_cycles / secs / 1e9
    peak = SIMDS * CLOCK_GHZ
    lanes = c["SQ_THREAD_CYCLES_VALU"]["mean"] / c["SQ_ACTIVE_INST_VALU"]["mean"]
    return {"achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "G busy SIMD-cycles/s",
            "frac": round(achieved / peak, 4),
            "valu_insts_per_launch": c["SQ_INSTS_VALU"]["mean"],
            "valu_insts_per_segment": round(c["SQ_INSTS_VALU"]["mean"] / segments, 3),
            "lanes_per_inst": round(lanes, 1),
            "useful_lane_frac": round(achieved / peak * lanes / 64.0, 4),
            "kernel_ms_under_pmc": c["_stamp"].get("kernel_ms_under_pmc"),
            "note": "the counter books 4 cycles per vector instruction; plain 32-bit VOP2 instructions issue in 2.3 "
                    "(profiles/r02_valu_cost.txt), so with several waves per SIMD the sum can pass the SIMD-cycles "
                    "available: a frac near or above 1 says the launch is at the VALU issue limit",
            "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU (own pass), %s, "
                      "source stamp %s" % (c["_path"], c["_stamp"]["source_sha"][:12])}


def hbm_traffic(c):
    """HBM bytes of one launch from the FETCH_SIZE / WRITE_SIZE passes, corrected as
    MI355X_MICROARCH.md prescribes: units of 1 KiB, reads x2 on gfx950 (128-B requests counted as 64 B)."""
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    return 2.0 * c["FETCH_SIZE"]["mean"] * 1024.0 + c["WRITE_SIZE"]["mean"] * 1024.0


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
    args = parse()
    if (args.gpus > 1 or args.force_dist) and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))  # before torch / HIP are even imported
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
    p = session.params
    W, H, spp = p.width, p.height, p.samples
    p.strip_rows, p.strip_count, p.strip_index = STRIP_ROWS, world, rank

    scene = rt.Scene(session, device=local)
    strips = importlib.import_module("racer-tracer_amd.strips")
    gatherer = strips.StripGather(H, W, STRIP_ROWS, world, rank, "cuda", dist, always_collective=args.force_dist)
    # when a collective runs the frame lives in the gather's staging buffer: no copies besides the strips
    frame = gatherer.frame() if gatherer.collective else torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream()
    kernel_ms, segments = [], []

    def step(record):
        # trace + resolve on torch's current stream, then the one collective of the path
        scene.render_frame_device(session.camera, p, frame.data_ptr(), stream.cuda_stream)
        gatherer.gather(frame)
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
        counters, why_not = pmc_summary(args.workload, workload, world)
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
            "roofline": {
                "bound": "valu",
                "achieved": None, "peak": round(SIMDS * CLOCK_GHZ, 1), "unit": "G busy SIMD-cycles/s", "frac": None,
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
            traffic = hbm_traffic(counters)
            if traffic is not None:
                # measured HBM bytes per launch over the live kernel duration
                out["roofline"]["traffic"] = round(traffic / (k_ms * 1e-3) / 1e9, 3)
                out["roofline"]["traffic_bytes_per_launch"] = traffic
                out["roofline"]["traffic_frac_of_hbm_peak"] = round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        else:
            out["roofline"]["counters_unavailable"] = why_not
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(session, args.cpu_seconds)
        if rehearsal:
            out["rehearsal"] = "N ranks on ONE GPU over gloo: functional check only"
        if args.force_dist:
            out["force_dist"] = "backend %s, world %d: process group initialised, strips gathered with dist.gather" % (
                dist.get_backend(), world)
        if rehearsal or args.force_dist:
            # the gathered frame must equal a single-rank render of the same frame
            p.strip_count = 0
            whole = torch.zeros_like(frame)
            scene.render_frame_device(session.camera, p, whole.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            out["rehearsal_frame_matches_single_rank"] = bool(torch.equal(whole, frame))
        print(json.dumps(out), flush=True)
    scene.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

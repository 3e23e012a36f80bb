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
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STRIP_ROWS = 8
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_SEGMENT_F64 = 192.0  # 96-B f64 ray record read + written per segment


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
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


def pmc_traffic(workload, spp, world):
    """HBM bytes per launch of the trace kernel, from the committed PMC summary
    of the SAME workload (profiles/pmc_traffic.json, produced by
    tools/gpu_pmc.sh + tools/pmc_to_traffic.py).  None when there is no
    measurement for this exact configuration."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if world != 1 or not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    return table.get("%s:%d" % (workload, spp))


def pmc_valu(workload, spp, world, kernel_ms, segments):
    """VALU view of the trace kernel from the committed PMC summary of the SAME workload
    (profiles/r01_c3_pmc_summary.json, rocprofv3 --pmc SQ_* in their own pass): the kernel is
    VALU-issue bound, so this - not the HBM fraction - says how busy the chip really is."""
    path = os.path.join(ROOT, "profiles", "r01_%s_pmc_summary.json" % workload)
    if world != 1 or spp != 1024 or not os.path.exists(path):
        return None
    with open(path) as f:
        summary = json.load(f)
    kernels = [k for k in summary if "k_trace_pool_f64" in k]
    if len(kernels) != 1 or "SQ_INSTS_VALU" not in summary[kernels[0]]:
        return None
    c = summary[kernels[0]]
    simd_cycles = kernel_ms * 1e-3 * 2.4e9 * 1024   # 256 CUs x 4 SIMDs at 2.4 GHz (MI355X_MICROARCH.md)
    return {"insts": c["SQ_INSTS_VALU"]["mean"],
            "insts_per_segment": round(c["SQ_INSTS_VALU"]["mean"] / segments, 3),
            "busy_frac": round(4.0 * c["SQ_ACTIVE_INST_VALU"]["mean"] / simd_cycles, 3),
            "lanes_per_inst": round(c["SQ_THREAD_CYCLES_VALU"]["mean"] / c["SQ_ACTIVE_INST_VALU"]["mean"], 1),
            "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU, profiles/r01_%s_pmc_summary.json" % workload}


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "--nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: there is no CPU fallback for the render path")
    # BENCH_REHEARSE_ON_ONE_GPU=1: every rank uses device 0 and the collective goes
    # over gloo (RCCL cannot put several ranks on one card).  A functional rehearsal of
    # the N > 1 path for a 1-GPU box; its numbers mean nothing.
    rehearsal = os.environ.get("BENCH_REHEARSE_ON_ONE_GPU") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
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
    gatherer = strips.StripGather(H, W, STRIP_ROWS, world, rank, "cuda", dist)
    # for N > 1 the frame lives in the gather's staging buffer: no copies besides the strips
    frame = gatherer.frame() if world > 1 else torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
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
    for _ in range(args.steps):
        step(True)
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

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = W * H * spp / (elapsed / args.steps) / 1e6
        seg_per_step = total_segments / args.steps
        k_ms = kernel_total_ms / args.steps
        achieved = BYTES_PER_SEGMENT_F64 * seg_per_step / (k_ms * 1e-3) / 1e9 / world  # GB/s per GPU
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
                                      % (world, ", RCCL gather to rank 0" if world > 1 else "")},
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None,
                "kernel": "k_trace_pool_f64",
                "kernel_ms": round(k_ms, 3),
                "segments_per_launch": seg_per_step / world,
                "bytes_per_segment": BYTES_PER_SEGMENT_F64,
                "algorithmic_bytes_per_launch": BYTES_PER_SEGMENT_F64 * seg_per_step / world,
                "gsegments_per_s": round(seg_per_step / (k_ms * 1e-3) / 1e9, 3),
                "note": "achieved = algorithmic ray-state bytes (192 B/segment, SURVEY 8d) / kernel time; "
                        "the kernel keeps ray state in registers, so its real HBM traffic is the per-chunk "
                        "framebuffer slices only and frac can exceed 1",
            },
        }
        pmc = pmc_traffic(args.workload, spp, world)
        if pmc is not None:
            # HBM bytes of one k_trace_pool_f64 launch from rocprofv3 PMC passes (tools/gpu_pmc.sh),
            # corrected as MI355X_MICROARCH.md prescribes, over the live kernel duration
            out["roofline"]["traffic"] = round(pmc["bytes_per_launch"] / (k_ms * 1e-3) / 1e9, 3)
            out["roofline"]["traffic_bytes_per_launch"] = pmc["bytes_per_launch"]
            out["roofline"]["traffic_source"] = pmc["source"]
        valu = pmc_valu(args.workload, spp, world, k_ms, seg_per_step)
        if valu is not None:
            out["roofline"]["valu"] = valu
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(session, args.cpu_seconds)
        if rehearsal:
            out["rehearsal"] = "N ranks on ONE GPU over gloo: functional check only"
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

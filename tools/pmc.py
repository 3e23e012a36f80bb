#!/usr/bin/env python3
"""rocprofv3 counter passes over the bench command, one pass per counter group.

    python3 tools/pmc.py <out.json> [--groups standard|all|"C1 C2" ...] -- [bench args, e.g. --workload c4]

Used two ways:
  * bench.py calls collect() itself before it touches the GPU (the roofline block of its JSON line is then LIVE:
    counters of this very tree on this very box);
  * tools/gpu_finalize_profiles.sh writes the per-workload summaries that are committed under profiles/.

Every pass is `rocprofv3 --pmc <group> -- python3 bench.py --pmc-child ...` (the program itself behind `--`, no
launcher in between; never combined with a trace option: MI355X_MICROARCH.md, HBM/rocprofv3 section).  The summary
holds, per kernel, the mean of every counter over the dispatches of the pass, plus a stamp: the hash of the kernel
sources (tools/source_stamp.py), the workload, and the kernel time each pass ran at.
"""
import collections
import csv
import json
import os
import shutil
import signal
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from source_stamp import kernel_source_sha, library_sha  # noqa: E402

# What bench.py's roofline block reads.  Four SQ counters per pass (the SQ has eight slots, derived counters take
# several); the TCC byte counters get passes of their own as the guide prescribes.
GROUPS = {
    "issue": "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_ACTIVE_INST_SCA",
    "waits": "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY",
    "f64": "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64",
    "int": "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU",
    "f32": "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32",
    "mem": "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR",
    "lds": "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS",
    "level": "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
}
STANDARD = ["issue", "waits", "f64", "int", "f32", "fetch", "write"]


def run_pass(counters, bench_args, keep_dir=None, timeout=240):
    """One rocprofv3 pass.  Returns ({kernel: {counter: [values]}}, bench_line or None, error text or None)."""
    work = tempfile.mkdtemp(prefix="rt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", work, "-o", "pmc", "--",
                                                     sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child"] + list(bench_args)
    # The pass runs in its own session (process group): if it has to be cut off, the whole group goes — rocprofv3 AND the
    # bench.py --pmc-child it started — so that no orphan keeps the GPU busy during the caller's timed steps.
    try:
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd="/tmp", start_new_session=True)
    except OSError as e:
        shutil.rmtree(work, ignore_errors=True)
        return None, None, "%s: %s" % (type(e).__name__, e)
    try:
        out, errtext = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired as e:
        try:
            os.killpg(proc.pid, signal.SIGKILL)  # pid == pgid: start_new_session
        except OSError:
            pass
        proc.communicate()
        shutil.rmtree(work, ignore_errors=True)
        return None, None, "%s: %s (process group killed)" % (type(e).__name__, e)
    run = subprocess.CompletedProcess(cmd, proc.returncode, out, errtext)
    bench = None
    for line in run.stdout.splitlines():
        if line.startswith("{"):
            try:
                bench = json.loads(line)
            except ValueError:
                pass
    found = None
    for base, _, files in os.walk(work):
        for f in sorted(files):
            if f.endswith("counter_collection.csv"):
                found = os.path.join(base, f)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    err = None
    if run.returncode != 0 or found is None:
        err = "rocprofv3 rc %d, %s; stderr tail: %s" % (run.returncode, "no counter_collection.csv" if found is None else "csv ok",
                                                         run.stderr[-400:].replace("\n", " | "))
    if found is not None:
        with open(found) as f:
            for r in csv.DictReader(f):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if keep_dir:
        os.makedirs(keep_dir, exist_ok=True)
        if found is not None:
            shutil.copy(found, os.path.join(keep_dir, "counter_collection.csv"))
        with open(os.path.join(keep_dir, "rocprof_stderr.log"), "w") as f:
            f.write(run.stderr)
    shutil.rmtree(work, ignore_errors=True)
    return agg, bench, err


def collect(bench_args, groups=None, keep_dir=None, log=None, budget_s=None, pass_timeout=240):
    """All passes -> summary dict (the format of profiles/rNN_<workload>_pmc_summary.json).
    budget_s: stop starting passes once this much wall time has gone (bench.py's live collection must not turn a
    misbehaving profiler into a bench run of half an hour)."""
    import time
    t_begin = time.time()
    names = list(groups or STANDARD)
    summary = {"_stamp": {"source_sha": kernel_source_sha(), "library_sha": library_sha(), "bench_args": " ".join(bench_args), "kernel_ms_under_pmc": [],
                          "groups": {}, "errors": []}}
    stamp = summary["_stamp"]
    for name in names:
        counters = GROUPS.get(name, name)
        timeout = pass_timeout
        if budget_s is not None:
            left = budget_s - (time.time() - t_begin)
            if left < 20:
                stamp["errors"].append("%s: skipped, the %d s budget of the live collection is spent" % (name, budget_s))
                continue
            timeout = min(timeout, left)
        agg, bench, err = run_pass(counters, bench_args, os.path.join(keep_dir, name.replace(" ", "_")) if keep_dir else None, timeout=timeout)
        if log:
            log("pmc pass %-6s %s" % (name, "ok" if err is None else err))
        if err is not None:
            stamp["errors"].append("%s: %s" % (name, err))
            if not stamp["groups"]:  # the very first pass failed (no GPU, no rocprofv3 ...): the others would too
                break
        if agg is None:
            continue
        stamp["groups"][name] = counters
        if bench:
            stamp["workload"] = bench["config"]["workload"]
            stamp["kernel_ms_under_pmc"].append(bench["roofline"]["kernel_ms"])
            stamp["segments_per_launch"] = bench["roofline"]["segments_per_launch"]
        for kernel, cs in agg.items():
            if "trace" not in kernel and "resolve" not in kernel and "post" not in kernel:
                continue
            for c, v in cs.items():
                summary.setdefault(kernel[:60], {})[c] = {"dispatches": len(v), "mean": sum(v) / len(v)}
    return summary


def main():
    argv = sys.argv[1:]
    if not argv:
        sys.exit(__doc__)
    out_path = argv.pop(0)
    groups, bench_args = [], []
    if "--" in argv:
        k = argv.index("--")
        argv, bench_args = argv[:k], argv[k + 1:]
    if argv and argv[0] == "--groups":
        groups = argv[1:]
    if groups == ["standard"] or not groups:
        groups = STANDARD
    elif groups == ["all"]:
        groups = list(GROUPS)
    summary = collect(bench_args, groups, keep_dir=os.path.splitext(out_path)[0] + "_passes", log=print)
    with open(out_path, "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    for k, cs in summary.items():
        if k == "_stamp":
            continue
        print(k)
        for c, v in sorted(cs.items()):
            print("   %-28s n=%d mean=%.6g" % (c, v["dispatches"], v["mean"]))
    if summary["_stamp"]["errors"]:
        print("errors:", summary["_stamp"]["errors"])
        sys.exit(1)


if __name__ == "__main__":
    main()

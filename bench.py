#!/usr/bin/env python3
"""bench.py -- images/sec of Engine::Forward() for YOLOv5s 640x640 fp32 on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process IS a rank;
started plainly, it spawns the N ranks itself as child processes (simpleinfer_amd/launch.py) before anything touches
the GPU and relays rank 0's JSON line.

A step is one Engine::Forward() over one batch of synthetic images (the reference's bench harness,
bench/bench_yolo.cpp:7-34, times exactly that) plus, for N > 1, the all-gather of the output slabs
([B/G, 25200, 85] -> [B, 25200, 85]): by default the direct fan-out over IPC-shared HBM of include/si_shard.h
(one device-to-device copy per peer, overlapped with the next step), RCCL's all_gather_into_tensor as the fallback
(`--gather rccl` forces it); the JSON line says which ran.  Default: per-GPU batch fixed (32) as N grows -- weak
scaling, global batch 32*N (N = 8 is BASELINE.json's "batch=256 sharded across 8 MI355X").  `--global-batch B` fixes the
total instead (strong scaling: B/N per GPU, BASELINE.json's "batch=32 at 1/2/4/8 GPUs").  Inputs are resident in HBM before
the timed region; weights are random-init (portable splitmix64 stream), data is synthetic.

The timed region is exactly K steps between two fences (barrier + device sync); it is repeated until `--min-time` seconds
of timed work have accumulated and `value` / `ms_per_step` are those of the MEDIAN window (`windows` carries count /
min / max / first).

One JSON line on stdout from rank 0.  Besides the contract's fields it carries
  roofline     -- the dominant kernel (the conv kernel TEMPLATE with the largest total time, all its instantiations):
                  algorithmic direct-conv FLOPs per launch / average launch duration, measured live with HIP
                  events on the engine's stream in an instrumented pass right after the timed region
                  (peak: 157.3 TFLOP/s fp32 MFMA); `instantiations` lists every instantiation under the exact name
                  rocprofv3 prints, with its launches, average duration, algorithmic bytes and the PMC `traffic`
                  recorded for it under profiles/; a Winograd-dominated config also carries frac_executed_mfma.
  cpu_baseline -- the CPU oracle (an unoptimised restatement of SimpleInfer's Eigen/highway path, kind "port") timed on
                  this box's host cores on bounded samples (~20 s, run BEFORE the GPU leg), rank 0, N = 1 only.
  host_io      -- the reference's host-tensor calling convention (Input borrows a host tensor, Extract returns host
                  memory): PCIe-inclusive, never `value`.
  app_pipeline -- test-yolo's flow end to end (u8 frames up, device letterbox, Forward, device NMS, boxes down), pipelined.
  gather / step_bound -- N > 1: where a step's time went (host waits in the gather, per-peer copy rate).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (images per Forward); weak scaling")
    ap.add_argument("--global-batch", type=int, default=0, help="total images per step over all GPUs (strong scaling): per-GPU batch = B / N")
    ap.add_argument("--gather", default="auto", choices=["auto", "p2p", "rccl", "both", "torch"],
                    help="N > 1 output all-gather, all behind the C-ABI of include/si_shard.h unless it says torch: p2p = direct IPC fan-out, "
                         "rccl = ncclAllGather through si_rccl_* (librccl.so by dlopen, no torch), auto = p2p with a collective fallback to "
                         "rccl, both = the timed region twice, p2p then rccl, in ONE run (`value` is p2p's, `gather_ab` carries both), "
                         "torch = torch.distributed all_gather_into_tensor (the round-1 path, kept for A/B)")
    ap.add_argument("--min-time", type=float, default=10.0, help="repeat the K-step timed window until this many seconds are accumulated (BASELINE.md section 3: >= 10 s)")
    ap.add_argument("--max-windows", type=int, default=200)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--model", default="yolov5s", choices=["yolov5s", "resnet18", "mobilenetv3"])
    ap.add_argument("--graph", type=int, default=0, help="replay Forward() as a hipGraph")
    ap.add_argument("--winograd", type=int, default=1, help="3x3 s1 convs: 0 implicit GEMM everywhere, 1 fused Winograd F(2,3) where faster (default), 2 fused Winograd F(4,3) on those layers")
    ap.add_argument("--fp16", type=int, default=0, help="1: fp16 storage / fp16 MFMA path (BASELINE.json configs[3]); the headline metric is fp32 (default 0)")
    ap.add_argument("--no-aux", action="store_true", help="skip the host-I/O and post-processing side measurements")
    ap.add_argument("--no-secondary", action="store_true", help="skip the bounded secondary configurations (fp16, ResNet18 b64, batch 8 / 4)")
    ap.add_argument("--secondary-time", type=float, default=1.0, help="seconds of timed windows per secondary configuration")
    ap.add_argument("--secondary-warmup", type=float, default=0.3, help="seconds of untimed forwards in front of them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=24, help="images in the batch-1 CPU baseline sample")
    ap.add_argument("--cpu-batch", type=int, default=16, help="batch of the batched CPU baseline samples (BASELINE.md section 3); the three CPU legs together are bounded to ~20 s")
    ap.add_argument("--cpu-threads", type=int, default=16, help="oracle threads (reference uses 16 intra-op)")
    ap.add_argument("--profile-passes", type=int, default=3)
    ap.add_argument("--layers", action="store_true", help="print the per-layer table to stderr")
    ap.add_argument("--engine-opt", action="append", default=[], metavar="KEY=VALUE",
                    help="extra Engine::SetOption for A/B runs (e.g. arena=0, fuse_upsample=0); recorded in config.engine_options")
    return ap.parse_args()


def workload_key(args, shape):
    """what a recorded traffic table is keyed on: the launches of a kernel instantiation are those of one model at one batch"""
    return "%s %dx%d %s batch %d" % (args.model, shape[1], shape[2], "fp16" if args.fp16 else "fp32", args.batch)


def build_model(mg, name, batch, size):
    if name == "yolov5s":
        return mg.build_yolov5s(batch, size), (batch, size, size, 3)
    sz = 224 if size == 640 else size
    if name == "mobilenetv3":
        return mg.build_mobilenetv3_small(batch, sz), (batch, sz, sz, 3)
    return mg.build_resnet18(batch, sz), (batch, sz, sz, 3)


def _physical_cores():
    """distinct (package, core) pairs of /proc/cpuinfo: hardware threads beyond that are SMT siblings"""
    try:
        cores, phys, core = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":", 1)[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        if cores:
            return len(cores)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_baseline(args, mg, td):
    """The oracle -- an UNOPTIMISED scalar-class restatement of the reference's Eigen/highway CPU path (kind "port"; plain C loops
    built with -ffp-contract=off, no cache blocking: it exists to check results, not to be fast) -- timed on this box's host cores
    on bounded samples of the same workload (~20 s for the three legs together).  A reported baseline, not a target:
      cpu_baseline           `--cpu-threads` threads (16 = the reference's intra-op pool, engine_impl.cpp:133), batch-1 forwards
      cpu_baseline_batched   the same threads, one forward of a batch-`--cpu-batch` model
      cpu_baseline_all_cores one thread per PHYSICAL core (threads bound to cores), the same batched forward"""
    # thread placement must be in the environment before libgomp initialises (the oracle is loaded lazily, below)
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    from oracle import orc
    name = _cpu_name()
    what = ("unoptimised CPU restatement of SimpleInfer's Eigen/highway path (Winograd F(2,3)+pack4 GEMM for 3x3 s1, im2col GEMM "
            "otherwise, unfused passes)")

    def sample(batch, forwards, threads, budget_s):
        orc.lib().orc_set_num_threads(threads)
        b, shape = build_model(mg, args.model, batch, args.size)
        pp, bp = os.path.join(td, "cpu%d.pnnx.param" % batch), os.path.join(td, "cpu%d.pnnx.bin" % batch)
        b.save(pp, bp)
        x = mg.synth_input(shape)
        t0 = time.perf_counter()
        done = 0
        for _ in range(forwards):
            orc.run_graph(pp, bp, {"0": x})
            done += 1
            if time.perf_counter() - t0 >= budget_s:
                break
        dt = time.perf_counter() - t0
        return {"value": round(batch * done / dt, 4), "unit": "images/sec", "cores": int(orc.lib().orc_num_threads()),
                "kind": "port",
                "sample": "%d forward(s) of %s %dx%d fp32 batch %d, %s, %.1f s, host %s"
                          % (done, args.model, shape[1], shape[2], batch, what, dt, name)}

    out = {"cpu_baseline": sample(1, max(args.cpu_images, 1), args.cpu_threads, 8.0)}
    if args.cpu_batch > 1:
        out["cpu_baseline_batched"] = sample(args.cpu_batch, 1, args.cpu_threads, 6.0)
        ncore = _physical_cores()
        if ncore > args.cpu_threads:
            out["cpu_baseline_all_cores"] = sample(min(args.cpu_batch, 8), 1, ncore, 6.0)
    return out


def _cpu_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip() + " x%d threads" % os.cpu_count()
    except OSError:
        pass
    return "unknown"


PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E


def _template_of(kernel):
    return kernel.split("<", 1)[0]


def _traffic_table(workload):
    """The PMC traffic table recorded for THIS workload (model, size, precision, per-GPU batch), or nothing: bytes per launch of a
    kernel instantiation mean something only for the launches they were measured on.  PMC counters cannot be collected from inside
    this process: the figures are the ones tools/run_traffic.sh recorded (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    passes) -- not a measurement of this run -- and every table names the workload and the commit it was recorded at."""
    import glob
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic*.json"))):
        try:
            table = json.load(open(tpath))
        except Exception:
            continue
        if table.get("_workload") == workload:
            return table, "profiles/%s (recorded at %s for %s)" % (os.path.basename(tpath), table.get("_recorded_at", "an earlier round"), workload)
    return {}, None


def _busy_table(workload):
    """The SQ-counter table recorded for THIS workload by tools/run_mfma_busy.sh (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ...,
    a pass of its own): the matrix pipe's busy fraction per kernel instantiation -- BASELINE.json's second metric ("conv MFMA util %") from the
    counter, not from TF/s / peak.  Like the traffic tables: recorded, not measured in this process; the table names its workload and commit."""
    import glob
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "mfma_busy*.json"))):
        try:
            table = json.load(open(tpath))
        except Exception:
            continue
        if table.get("_workload") == workload:
            return table, "profiles/%s (rocprofv3 --pmc, recorded at %s for %s)" % (os.path.basename(tpath), table.get("_recorded_at", "an earlier commit"), workload)
    return {}, None


def _busy_of(table, name, field="mfma_busy"):
    rec = table.get(name)
    if isinstance(rec, dict):
        return rec.get(field)
    stem = name.rstrip(">")
    hits = [v for k, v in table.items() if k.startswith(stem) and isinstance(v, dict) and v.get(field) is not None]
    n = sum(v.get("launches", 0) for v in hits)
    return round(sum(v[field] * v.get("launches", 0) for v in hits) / n, 4) if n else None


def _traffic_of(table, name):
    rec = table.get(name)
    if rec is None:
        # a profile name may drop trailing template arguments the profiler prints: launch-weighted mean over the
        # instantiations that share the prefix
        stem = name.rstrip(">")
        hits = [v for k, v in table.items() if k.startswith(stem) and isinstance(v, dict)]
        n = sum(v.get("launches", 0) for v in hits)
        if n:
            return round(sum(v["hbm_bytes_per_launch"] * v.get("launches", 0) for v in hits) / n)
        return None
    return rec.get("hbm_bytes_per_launch") if isinstance(rec, dict) else rec


WINOGRAD_MULT_REDUCTION = {"conv_wino23_kernel": 2.25, "conv_wino43_kernel": 4.0, "conv_wino23s_kernel": 2.25}   # direct multiplies per executed multiply


def roofline_from_profile(passes, fp16=False, workload=None, busy_workload=None):
    """passes: list of per-layer profile lists (same schedule).  The dominant KERNEL is the conv kernel template with the
    largest total time (all its instantiations: one source kernel whose tile / MFMA shape follows the launch size); it is
    priced as SUM of algorithmic work / SUM of launch durations, and every instantiation -- the names are exactly what
    rocprofv3 --kernel-trace prints -- is listed beside it with its own launches, average duration, algorithmic bytes and
    PMC traffic, so the rocprof summary under profiles/ can be checked row by row.
    fp32: priced against the fp32 MFMA peak with direct-conv FLOPs; a Winograd kernel executes 2.25x (4x)
    fewer multiplies than it is credited with, so its line also carries frac_executed_mfma = frac / 2.25.
    fp16: every YOLOv5s layer is bound by memory, so the kernel is priced against HBM with its algorithmic bytes."""
    inst = {}
    peak_tf = PEAK_F16_MFMA_TFLOPS if fp16 else PEAK_FP32_MFMA_TFLOPS
    for layers in passes:
        for L in layers:
            if not L["kernel"].startswith("conv_") or L["flops"] <= 0 or L["type"] == "models.yolo.Detect":
                continue
            a = inst.setdefault(L["kernel"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "t_flop": 0.0, "t_byte": 0.0, "t_bound": 0.0})
            a["ms"] += L["ms"]
            a["flops"] += L["flops"]
            a["bytes"] += L["bytes"]
            a["launches"] += 1
            tf = L["flops"] / (peak_tf * 1e12)
            tb = L["bytes"] / (PEAK_HBM_GBS * 1e9)
            a["t_flop"] += tf
            a["t_byte"] += tb
            a["t_bound"] += max(tf, tb)
    if not inst:
        return None, inst
    tmpl = {}
    for k, v in inst.items():
        t = tmpl.setdefault(_template_of(k), {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "t_flop": 0.0, "t_byte": 0.0, "t_bound": 0.0})
        for f in ("ms", "flops", "bytes", "launches", "t_flop", "t_byte", "t_bound"):
            t[f] += v[f]
    name, a = max(tmpl.items(), key=lambda kv: kv[1]["ms"])
    npass = max(len(passes), 1)
    avg_ms = a["ms"] / a["launches"]
    flops_per_launch = a["flops"] / a["launches"]
    bytes_per_launch = a["bytes"] / a["launches"]
    achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
    table, traffic_source = _traffic_table(workload)
    btable, busy_source = _busy_table(busy_workload or workload)
    rows, tsum, tn, bsum, bn = [], 0.0, 0, 0.0, 0.0
    for k, v in sorted(inst.items(), key=lambda kv: -kv[1]["ms"]):
        if _template_of(k) != name:
            continue
        tr = _traffic_of(table, k)
        if tr is not None:
            tsum += tr * v["launches"]
            tn += v["launches"]
        ims = v["ms"] / v["launches"]
        busy = _busy_of(btable, k)
        if busy is not None:   # time-weighted over the instantiations: the busy fraction of the template's own kernel time
            bsum += busy * v["ms"]
            bn += v["ms"]
        row = {"kernel": k, "launches_per_step": v["launches"] // npass, "avg_launch_ms": round(ims, 4),
               "algorithmic_bytes": round(v["bytes"] / v["launches"]), "traffic": tr,
               "traffic_over_algorithmic": round(tr / (v["bytes"] / v["launches"]), 3) if tr else None,
               "mfma_busy": busy, "valu_per_mfma": _busy_of(btable, k, "valu_per_mfma")}
        row["gbs"] = round(v["bytes"] / v["launches"] / (ims * 1e-3) / 1e9, 1)
        row["tflops"] = round(v["flops"] / v["launches"] / (ims * 1e-3) / 1e12, 2)
        if fp16:
            row["frac"] = round(row["gbs"] / PEAK_HBM_GBS, 4)
        else:
            row["frac"] = round(row["tflops"] / PEAK_FP32_MFMA_TFLOPS, 4)
        # which roofline this instantiation's launches sit under: the longer of the time its algorithmic FLOPs
        # take at the matrix peak and the time its algorithmic bytes take at the HBM peak -- summed launch by launch, because
        # one instantiation serves layers on both sides of the ridge -- and the fraction of THAT bound it reaches
        row["bound"] = "mfma" if v["t_flop"] >= v["t_byte"] else "hbm"
        row["frac_of_bound"] = round(v["t_bound"] / (v["ms"] * 1e-3), 4)
        row["share_of_template_time"] = round(v["ms"] / a["ms"], 4)
        rows.append(row)
    # launch-weighted over the instantiations that have a recorded figure (None when none has)
    traffic = round(tsum / tn) if tn else None
    label = name + (" (all %d instantiations)" % len(rows) if len(rows) > 1 else "")
    if len(rows) == 1:
        label = rows[0]["kernel"]
    common = {"kernel": label,
              # the single largest instantiation by time (rows are sorted by it), named beside the family
              "largest_instantiation": {"kernel": rows[0]["kernel"], "share_of_template_time": rows[0]["share_of_template_time"],
                                        "bound": rows[0]["bound"], "frac_of_bound": rows[0]["frac_of_bound"],
                                        "tflops": rows[0]["tflops"], "gbs": rows[0]["gbs"]},
              # every launch against the roofline it sits under: sum of max(flop-time, byte-time) / sum of durations
              "frac_bound_aware": round(a["t_bound"] / (a["ms"] * 1e-3), 4),
              "traffic": traffic, "traffic_source": traffic_source if traffic is not None else None,
              # BASELINE.json's second metric from the COUNTER: SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32), time-weighted over the
              # template's instantiations (tools/run_mfma_busy.sh); beside it, `frac` is TF/s / peak
              "mfma_busy": round(bsum / bn, 4) if bn else None, "mfma_busy_source": busy_source if bn else None,
              "algorithmic_bytes": round(bytes_per_launch), "launches_per_step": a["launches"] // npass,
              "avg_launch_ms": round(avg_ms, 4), "instantiations": rows}
    if fp16:
        gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                "mbytes_per_launch": round(bytes_per_launch / 1e6, 2), "tflops": round(achieved, 1),
                "frac_of_f16_mfma_peak": round(achieved / PEAK_F16_MFMA_TFLOPS, 4)}
    else:
        roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "gflop_per_launch": round(flops_per_launch / 1e9, 3)}
        red = WINOGRAD_MULT_REDUCTION.get(name)
        if red:
            # `achieved` credits the kernel with DIRECT-convolution FLOPs, so frac can exceed 1; the matrix
            # pipe's own utilisation is the executed work: direct / 2.25 for F(2,3), / 4 for F(4,3)
            roof["frac_executed_mfma"] = round(achieved / red / PEAK_FP32_MFMA_TFLOPS, 4)
            roof["note"] = "Winograd: frac counts direct-conv FLOPs; frac_executed_mfma = frac / %.2f is the matrix pipe's share" % red
    roof.update(common)
    return roof, inst


def secondary_measurements(args, si, hipops, H, mg, td, dev):
    """BASELINE.json's other single-GPU configurations and the per-GPU batches of the headline metric's 4 / 8-GPU points, in the
    driver's line: YOLOv5s fp16 batch 32 (configs[3]), ResNet18 224x224 fp32 batch 64 (configs[2]), YOLOv5s
    fp32 batch 8 and batch 4 (batch 32 over 4 / 8 GPUs).  Each: device-resident input, >= --secondary-time seconds of ~80 ms
    windows after --secondary-warmup seconds of forwards, median window.  `value` of the line stays the fp32 batch-32 headline; these are bounded (~6 s of
    timing) side figures with the roofline each one sits under: fp32 nets against the conv-GEMM MFMA ceiling (direct-conv
    FLOPs / 157.3 TF/s), the fp16 net against HBM (algorithmic bytes of its conv layers / 8 TB/s) as BASELINE.md prices it."""
    out = {}
    cases = [("yolov5s_fp16_b32", "yolov5s", 32, 640, 1, {}), ("resnet18_fp32_b64", "resnet18", 64, 224, 0, {}),
             ("yolov5s_fp32_b8", "yolov5s", 8, 640, 0, {}), ("yolov5s_fp32_b4", "yolov5s", 4, 640, 0, {}),
             # opt-in (round 5): fp32 tensors, the K-heavy convs contracted from three fp16 MFMA products per fp32
             # product (engine option f32_split; conv_split3.hip).  Beside the headline, never instead of it: `value` stays true fp32.
             ("yolov5s_f32split_b32", "yolov5s", 32, 640, 0, {"f32_split": 1}),
             # ... at the per-GPU batch of the 8-GPU strong-scaling point: the split path must not LOSE where the launches are small
             ("yolov5s_f32split_b4", "yolov5s", 4, 640, 0, {"f32_split": 1})]
    for key, model, batch, size, fp16, opts in cases:
        try:
            builder, shape = build_model(mg, model, batch, size)
            pp, bp = os.path.join(td, key + ".param"), os.path.join(td, key + ".bin")
            builder.save(pp, bp)
            flops = mg.conv_flops(builder)
            e = si.Engine(device=dev, outputs_to_host=0, graph=args.graph, winograd=args.winograd, fp16=fp16, **opts)
            e.load_model(pp, bp)
            dx = hipops.DeviceBuffer.from_numpy(mg.synth_input(shape, seed=1))
            e.input_device(e.input_names()[0], dx.ptr)
            # warm-up by TIME (the clocks have followed the PCIe-bound side measurements down; three forwards of a 1 ms step do not
            # bring them back), then windows of about the headline leg's length (20 steps x 4.2 ms), at least 10 steps each
            t0 = time.perf_counter()
            nw = 0
            while nw < 3 or time.perf_counter() - t0 < args.secondary_warmup:
                e.forward()
                nw += 1
                if nw % 8 == 0:
                    H.si_hip_device_sync()
            H.si_hip_device_sync()
            est = (time.perf_counter() - t0) / nw
            steps, ws, total = max(10, min(100, int(0.08 / max(est, 1e-4)))), [], 0.0
            while total < args.secondary_time:
                t0 = time.perf_counter()
                for _ in range(steps):
                    e.forward()
                H.si_hip_device_sync()
                ws.append(time.perf_counter() - t0)
                total += ws[-1]
            ws.sort()
            dt = ws[len(ws) // 2] / steps
            rec = {"value": round(batch / dt, 1), "unit": "images/sec", "ms_per_step": round(dt * 1e3, 3), "windows": len(ws), "steps_per_window": steps,
                   "workload": "%s %dx%d %s batch %d" % (model, shape[1], shape[2], "fp16" if fp16 else "fp32", batch)}
            # every entry carries its dominant kernel the way the headline does: template, largest instantiation with
            # the roofline IT sits under, PMC traffic when a table was recorded for this workload; a Winograd kernel is credited with
            # direct-conv FLOPs, so wherever such a figure appears its executed-work twin stands beside it
            layers = e.profile()
            roof, _ = roofline_from_profile([layers], fp16=bool(fp16), workload=rec["workload"],
                                            busy_workload=rec["workload"] + (" f32_split=1" if opts.get("f32_split") else ""))
            convs = [L for L in layers if L["kernel"].startswith("conv_") and L["flops"] > 0]
            executed = sum(L["flops"] / WINOGRAD_MULT_REDUCTION.get(_template_of(L["kernel"]), 1.0) for L in convs)
            credited = sum(L["flops"] for L in convs)
            if fp16:
                cbytes = sum(L["bytes"] for L in layers if L["kernel"].startswith("conv_"))
                cms = sum(L["ms"] for L in layers if L["kernel"].startswith("conv_"))
                rec.update({"bound": "hbm", "frac": round(cbytes / (cms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if cms > 0 else None,
                            "frac_is": "algorithmic bytes of the conv launches / their event-timed durations / 8 TB/s",
                            "frac_of_f16_mfma_ceiling": round(batch / dt / (PEAK_F16_MFMA_TFLOPS * 1e12 / (flops / batch)), 4),
                            "kernels": sorted({L["kernel"].split("<")[0] for L in layers})})
            else:
                frac = batch / dt / (PEAK_FP32_MFMA_TFLOPS * 1e12 / (flops / batch))
                rec.update({"bound": "mfma", "frac": round(frac, 4),
                            "frac_is": "images/s / (157.3 TF/s / direct-conv FLOPs per image): DIRECT-CONV CREDIT -- Winograd layers execute "
                                       "2.25x fewer multiplies than they are credited with, so this can exceed 1",
                            "frac_executed_mfma": round(frac * executed / credited, 4) if credited > 0 else None,
                            "frac_executed_mfma_is": "the same with every Winograd F(2,3) layer counted at the multiplies it executes (direct / 2.25): "
                                                     "the matrix pipe's real share"})
            if roof:
                dom = {"kernel": roof["kernel"], "launches_per_step": roof["launches_per_step"], "avg_launch_ms": roof["avg_launch_ms"],
                       "bound": roof["bound"], "frac": roof["frac"], "frac_bound_aware": roof["frac_bound_aware"],
                       "largest_instantiation": roof["largest_instantiation"], "traffic": roof["traffic"],
                       "mfma_busy": roof.get("mfma_busy"), "mfma_busy_source": roof.get("mfma_busy_source"),
                       "algorithmic_bytes": roof["algorithmic_bytes"], "traffic_source": roof["traffic_source"]}
                if roof.get("traffic"):
                    dom["traffic_over_algorithmic"] = round(roof["traffic"] / roof["algorithmic_bytes"], 3)
                if "frac_executed_mfma" in roof:
                    dom["frac_is"] = "direct-conv credit"
                    dom["frac_executed_mfma"] = roof["frac_executed_mfma"]
                rec["dominant_kernel"] = dom
            if opts.get("f32_split"):
                # this entry's arithmetic and its roofline: the split kernel executes THREE fp16 MFMA products per credited multiply, so
                # it is priced on executed fp16 MFMA FLOPs against the fp16 peak (a fraction of the fp32 peak above 1 is not a roofline
                # fraction); the net-level `frac` above stays direct-conv credit against the fp32 peak, labelled as such
                sp = [L for L in convs if L["kernel"].startswith("conv_split3")]
                sms, sfl = sum(L["ms"] for L in sp), sum(L["flops"] for L in sp)
                rec["dtype"] = "f32 (3 x f16 split products, fp32 accumulate)"
                rec["frac_is"] = "images/s / (157.3 TF/s / direct-conv FLOPs per image): DIRECT-CONV CREDIT against the FP32 peak; the split layers run on the fp16 pipe -- see split_kernel"
                ws = [L for L in convs if L["kernel"].startswith("conv_wino23s")]
                wms, wfl = sum(L["ms"] for L in ws), sum(L["flops"] for L in ws)
                if ws and wms > 0:
                    # the Winograd layers on the split form of the fused Winograd kernel: 16 / 36 of the direct multiplies, three fp16 products each
                    rec["split_winograd_kernel"] = {"kernel": "conv_wino23s_kernel", "launches_per_step": len(ws), "ms_per_step": round(wms, 3),
                                                    "share_of_conv_time": round(wms / sum(L["ms"] for L in convs), 3),
                                                    "tflops_direct_equivalent": round(wfl / (wms * 1e-3) / 1e12, 1),
                                                    "executed_f16_mfma_tflops": round(3.0 * wfl / 2.25 / (wms * 1e-3) / 1e12, 1), "bound": "vector issue (transform + split), not a pipe",
                                                    "peak": PEAK_F16_MFMA_TFLOPS, "frac": round(3.0 * wfl / 2.25 / (wms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
                                                    "frac_is": "3 x (direct-conv FLOPs / 2.25) of the Winograd layers / their event-timed durations / 2500 TF/s"}
                st = [L for L in convs if L["kernel"].startswith("conv_stem_split")]
                if st and st[0]["ms"] > 0:
                    # the RGB stem on the split form of the fp16 stem kernel: 108-deep contraction, 420 MB of fp32 activations out -- HBM-bound
                    gbs = st[0]["bytes"] / (st[0]["ms"] * 1e-3) / 1e9
                    rec["split_stem_kernel"] = {"kernel": "conv_stem_split_f32_kernel", "ms_per_step": round(st[0]["ms"], 3), "bound": "hbm",
                                                "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                                                "frac_is": "(image + activations + weights, once each) / event-timed duration / 8 TB/s"}
                bt, bsrc = _busy_table(rec["workload"] + " f32_split=1")
                rec["guard"] = {"split_reruns": e.schedule().get("split_reruns"), "split_demoted": e.schedule().get("split_demoted"),
                                "note": "range guard (include/si_hip.h): a layer whose operands leave fp16's range goes back to the true-fp32 kernels and "
                                        "the step is re-run; synthetic U[0,1) images never trip it"}
                if sp and sms > 0:
                    rec["split_kernel_mfma_busy"] = {"conv_split3_f32_kernel": _busy_of(bt, "conv_split3_f32_kernel<"),   # (launch-weighted over its instantiations)
                                                     "conv_wino23s_kernel": _busy_of(bt, "conv_wino23s_kernel<"),
                                                     "conv_stem_split_f32_kernel": _busy_of(bt, "conv_stem_split_f32_kernel<"), "source": bsrc}
                    rec["split_kernel"] = {"kernel": "conv_split3_f32_kernel", "launches_per_step": len(sp), "ms_per_step": round(sms, 3),
                                           "share_of_conv_time": round(sms / sum(L["ms"] for L in convs), 3),
                                           "tflops_direct_equivalent": round(sfl / (sms * 1e-3) / 1e12, 1),
                                           "executed_f16_mfma_tflops": round(3.0 * sfl / (sms * 1e-3) / 1e12, 1), "bound": "mfma (fp16)",
                                           "peak": PEAK_F16_MFMA_TFLOPS, "frac": round(3.0 * sfl / (sms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
                                           "frac_is": "3 x direct-conv FLOPs of the split layers / their event-timed durations / 2500 TF/s (dense fp16 MFMA)"}
            out[key] = rec
            e.release()
            dx.free()
        except Exception as ex:  # noqa: BLE001 -- a side figure must never cost the headline line
            out[key] = {"error": str(ex)}
    return out


def aux_measurements(args, si, hipops, H, e, oname, oshape, pp, bp, dev, x, extra_opts=None):
    """Reported beside the headline, never part of `value`: (1) the PCIe-inclusive rate -- the same forward with the
    input handed over as a host buffer (re-uploaded on every Forward, as the reference's Input() contract requires)
    and the output slab copied back to host memory; (2) the device-side detection post-processing
    (si_hip_yolo_postprocess_f32, test_yolo.cpp:337-428) on the resident Detect output."""
    import ctypes as C
    aux = {}
    steps = max(3, min(args.steps, 5))
    # pin_inputs: `x` stays alive and mapped until e2.release() below, which is what the opt-in asks of the caller
    e2 = si.Engine(device=dev, outputs_to_host=1, graph=args.graph, winograd=args.winograd, fp16=args.fp16, pin_inputs=1, **(extra_opts or {}))
    e2.load_model(pp, bp)
    e2.input(e2.input_names()[0], x)
    for _ in range(3):   # the first Forward builds the sliced pipeline, the second pins the borrowed input buffer in place
        e2.forward()
    H.si_hip_device_sync()
    hsteps = max(steps, 12)
    t0 = time.perf_counter()
    for _ in range(hsteps):
        e2.forward()
    H.si_hip_device_sync()
    dt = time.perf_counter() - t0
    aux["host_io"] = {"value": round(args.batch * hsteps / dt, 2), "unit": "images/sec", "ms_per_step": round(dt / hsteps * 1e3, 3),
                      "note": "the reference's calling convention (bench/bench_yolo.cpp:20-28): Input() borrows a HOST tensor that is "
                              "uploaded on every Forward(), Extract() returns host memory (PCIe-inclusive: 4.9 MB up + 8.6 MB down per "
                              "image); one synchronous Forward() pipelines batch slices over an upload stream, a compute stream and a "
                              "download stream (engine option host_slices; the borrowed input buffer pinned in place: pin_inputs=1, opt-in); "
                              "not `value`"}
    e2.release()
    if args.model == "yolov5s" and len(oshape) == 3:
        n, rows, ne = oshape
        optr, on_dev = e.extract_ptr(oname)
        max_det = 300
        wsb = H.si_hip_yolo_postprocess_workspace_bytes(n, rows, ne)
        ws, dets, cnt = hipops.DeviceBuffer(wsb), hipops.DeviceBuffer(n * max_det * 6 * 4), hipops.DeviceBuffer(n * 4)
        ev0, ev1 = C.c_void_p(), C.c_void_p()
        H.si_hip_event_create(C.byref(ev0)); H.si_hip_event_create(C.byref(ev1))
        ms = C.c_float()
        def timed(ptr):
            times = []
            for _ in range(steps + 1):
                H.si_hip_event_record(ev0, None)
                rc = H.si_hip_yolo_postprocess_f32(ptr, n, rows, ne, 0.25, 0.45, 0, None, dets.ptr, cnt.ptr, max_det, ws.ptr, wsb, None)
                H.si_hip_event_record(ev1, None)
                H.si_hip_event_sync(ev1)
                H.si_hip_event_elapsed_ms(ev0, ev1, C.byref(ms))
                times.append(ms.value)
                if rc != 0:
                    raise RuntimeError("si_hip_yolo_postprocess_f32 rc=%d" % rc)
            return round(float(np.mean(times[1:])), 3), round(float(cnt.to_numpy((n,), np.int32).mean()), 1)

        # (a) detector-like predictions: a few objects per image, ~750 rows above the confidence threshold
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from util import synthetic_predictions
        dsyn = hipops.DeviceBuffer.from_numpy(synthetic_predictions(3, n, rows, nc=ne - 5, n_gt=8, hot_frac=0.03))
        ms_a, kept_a = timed(dsyn.ptr)
        # (b) this run's Detect output: random-init weights put ~all 25200 rows above the threshold, mostly in one class
        ms_b, kept_b = timed(optr)
        H.si_hip_event_destroy(ev0); H.si_hip_event_destroy(ev1)
        aux["postprocess"] = {"ms_per_batch": ms_a, "boxes_kept_per_image": kept_a,
                              "ms_per_batch_degenerate": ms_b, "boxes_kept_per_image_degenerate": kept_b,
                              "bytes_out_per_image": max_det * 24 + 4, "bytes_in_per_image": rows * ne * 4,
                              "note": "confidence filter + sort + per-class NMS on the device (test_yolo.cpp:337-428) over a resident "
                                      "[n,25200,85] slab: detector-like synthetic predictions; 'degenerate' = this run's random-init "
                                      "network output, where nearly every row passes the 0.25 filter in one class"}
    return aux


def app_pipeline(args, si, hipops, H, e, iname, oname, oshape, rate_resident):
    """The application around Forward() in the reference's test-yolo (test/test_yolo/test_yolo.cpp:299-438) as a pipelined
    stream of batches, EVERY stage on the device (round 4: cv::resize too): per batch the host holds 720 x 1280 u8 BGR camera
    frames; they are uploaded (u8: 88 MB per 32 images) on a copy stream into one of TWO device buffers while the previous batch
    computes; on the engine's stream: bilinear resize + letterbox (pad / BGR->RGB / cast / divide by 255, :194-259) in one launch
    into the engine's input, Forward(), confidence filter + sort + per-class NMS
    (:337-428), and the boxes (7 KB per image) come back.  Reported: images/sec end to end and its ratio to the device-resident
    Forward() rate of the same run.  Random-init weights put every row near confidence 0.25, so the confidence threshold is set
    where 3 % of the rows pass (~750 candidates per image, what a trained detector yields), and says so."""
    import ctypes as C
    n, size = args.batch, args.size
    rows, ne = oshape[1], oshape[2]
    # round 4: the host holds CAMERA frames (720 x 1280 BGR u8, 2.8 MB each) and the whole of PreProcess -- cv::resize included --
    # runs on the device (si_hip_resize_letterbox_batch_u8_f32: bilinear resize + letterbox in one launch)
    cam_h, cam_w = 720, 1280
    hr, wr = cam_h, cam_w
    img_bytes = hr * wr * 3
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (2, n, hr, wr, 3), dtype=np.uint8)
    pinned, dev_u8, dev_in = [], [], []
    for k in range(2):
        hp = C.c_void_p()
        assert H.si_hip_host_alloc(C.byref(hp), n * img_bytes) == 0
        C.memmove(hp, frames[k].ctypes.data, n * img_bytes)
        pinned.append(hp)
        dev_u8.append(hipops.DeviceBuffer(n * img_bytes))
        dev_in.append(hipops.DeviceBuffer(n * size * size * 3 * 4))
    max_det = 300
    wsb = H.si_hip_yolo_postprocess_workspace_bytes(n, rows, ne)
    ws, dets, cnt = hipops.DeviceBuffer(wsb), hipops.DeviceBuffer(n * max_det * 6 * 4), hipops.DeviceBuffer(n * 4)
    hdets = [C.c_void_p(), C.c_void_p()]
    for k in range(2):
        assert H.si_hip_host_alloc(C.byref(hdets[k]), n * max_det * 6 * 4 + n * 4) == 0
    copy_stream = C.c_void_p()
    assert H.si_hip_stream_create(C.byref(copy_stream)) == 0
    ev_up = [C.c_void_p(), C.c_void_p()]
    ev_used = [C.c_void_p(), C.c_void_p()]
    ev_done = [C.c_void_p(), C.c_void_p()]
    for evs in (ev_up, ev_used, ev_done):
        for k in range(2):
            H.si_hip_event_create(C.byref(evs[k]))
    es = e.stream()
    optr, _ = e.extract_ptr(oname)

    def upload(k):
        H.si_hip_stream_wait_event(copy_stream, ev_used[k])          # the letterbox of two batches ago has read this buffer
        H.si_hip_memcpy_h2d(dev_u8[k].ptr, pinned[k], n * img_bytes, copy_stream)
        H.si_hip_event_record(ev_up[k], copy_stream)

    def compute(k, thr):
        H.si_hip_stream_wait_event(es, ev_up[k])
        rc = H.si_hip_resize_letterbox_batch_u8_f32(dev_u8[k].ptr, n, img_bytes, cam_h, cam_w, dev_in[k].ptr, size, size, es)
        if rc != 0:
            raise RuntimeError("si_hip_resize_letterbox_batch_u8_f32 rc=%d" % rc)
        H.si_hip_event_record(ev_used[k], es)
        e.input_device(iname, dev_in[k].ptr)
        e.forward_async()
        rc = H.si_hip_yolo_postprocess_f32(optr, n, rows, ne, thr, 0.45, 0, None, dets.ptr, cnt.ptr, max_det, ws.ptr, wsb, es)
        if rc != 0:
            raise RuntimeError("si_hip_yolo_postprocess_f32 rc=%d" % rc)
        H.si_hip_memcpy_d2h(hdets[k], dets.ptr, n * max_det * 6 * 4, es)
        H.si_hip_memcpy_d2h(C.c_void_p(hdets[k].value + n * max_det * 6 * 4), cnt.ptr, n * 4, es)
        H.si_hip_event_record(ev_done[k], es)

    for k in range(2):
        H.si_hip_event_record(ev_used[k], es)
    # threshold where 3 % of the rows of this network's output pass
    upload(0)
    compute(0, 2.0)
    e.sync()
    pred = hipops.DeviceBuffer.view(optr, n * rows * ne * 4).to_numpy((n, rows, ne))
    conf = pred[..., 4] * pred[..., 5:].max(axis=-1)
    thr = float(np.quantile(conf, 0.97))
    upload(1)
    for it in range(4):            # warm-up
        compute((it + 1) % 2, thr)
        upload(it % 2)
    e.sync()
    steps = max(args.steps, 20)
    t0 = time.perf_counter()
    for it in range(steps):
        k = (it + 1) % 2
        compute(k, thr)
        upload(1 - k)              # the next batch goes up while this one computes
        if it > 0:
            H.si_hip_event_sync(ev_done[1 - k])   # the previous batch's boxes are on the host
    e.sync()
    dt = time.perf_counter() - t0
    kept = np.ctypeslib.as_array(C.cast(C.c_void_p(hdets[k].value + n * max_det * 6 * 4), C.POINTER(C.c_int32)), shape=(n,)).mean()
    for k in range(2):
        H.si_hip_host_free(pinned[k]); H.si_hip_host_free(hdets[k])
        dev_u8[k].free(); dev_in[k].free()
    for evs in (ev_up, ev_used, ev_done):
        for k in range(2):
            H.si_hip_event_destroy(evs[k])
    H.si_hip_stream_destroy(copy_stream)
    value = n * steps / dt
    return {"value": round(value, 2), "unit": "images/sec", "ms_per_batch": round(dt / steps * 1e3, 3),
            "ratio_to_device_resident": round(value / rate_resident, 4), "boxes_kept_per_image": round(float(kept), 1),
            "confidence_threshold": round(thr, 4), "bytes_up_per_image": img_bytes, "bytes_down_per_image": max_det * 24 + 4,
            "note": "test-yolo's flow (test_yolo.cpp:299-438) pipelined, every stage on the device: %dx%d u8 BGR camera frames uploaded "
                    "double-buffered on a copy stream, bilinear resize + letterbox to %dx%d in one launch (PreProcess incl. cv::resize) -> "
                    "Forward -> device filter / sort / NMS, boxes downloaded; threshold at the 97th percentile of this random-init "
                    "network's confidences (3 %% of rows pass, as with a trained detector); not `value`" % (cam_h, cam_w, size, size)}


def self_launch(args):
    """`python bench.py --gpus N` with no launcher in the environment: become the launcher.  Nothing in this process has
    touched (or will touch) the GPU -- the ranks are children."""
    from simpleinfer_amd import launch
    code, out = launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if code == 0 and lines:
        print(lines[-1], flush=True)
    elif code == 0:
        code = 1
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
    sys.exit(code)


def main():
    args = parse()
    from simpleinfer_amd import launch
    if args.gpus > 1 and not launch.launched_by_a_launcher():
        self_launch(args)   # does not return

    import simpleinfer_amd as si
    from simpleinfer_amd import distributed as sd, hipops, shard, _native

    rank, world, local_rank = sd.env_rank_world()
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.global_batch:
        if args.global_batch % world:
            sys.exit("bench.py: --global-batch %d is not divisible by %d GPUs" % (args.global_batch, world))
        args.batch = args.global_batch // world
    H = _native.hip()
    ndev = si.device_count()
    if ndev <= 0:
        sys.exit("bench.py: no HIP device (the product has no CPU fallback)")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    # SI_BENCH_SHARE_DEVICE=1: ranks may share a device (exercises the multi-rank path, incl. the IPC gather, on a 1-GPU box)
    share = os.environ.get("SI_BENCH_SHARE_DEVICE") == "1"
    if os.environ.get("SI_LAUNCH_PIN_VISIBLE") == "1":
        local_world = 1   # every rank sees only its own GPU (simpleinfer_amd/launch.py)
    if local_world > ndev and not share:
        sys.exit("bench.py: need %d HIP devices for --gpus %d, found %d" % (local_world, world, ndev))
    dev = local_rank % ndev

    # the contract is ONE JSON line on stdout: RCCL prints a version banner to stdout when its communicator is created,
    # so everything below runs with fd 1 pointed at stderr and the real stdout is restored for the final line only
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    H.si_hip_set_device(dev)

    # SI_BENCH_FORCE_DIST=1 runs the N>1 code path even at world size 1, so it can be exercised on a 1-GPU box
    use_dist = world > 1 or os.environ.get("SI_BENCH_FORCE_DIST") == "1"
    group = None       # node-local rank group (p2p mode)
    dist = torch = None
    gather_mode = None
    gather_note = None
    if use_dist:
        want = args.gather
        if want != "torch":
            try:
                group = shard.NodeGroup(shard.default_group_name(), rank, world, timeout_s=120.0)
                gather_mode = "rccl" if want == "rccl" else "p2p"
            except Exception as ex:  # every rank times out together when the rendezvous cannot form
                if want != "auto":
                    raise
                gather_note = "node group unavailable (%s)" % ex
        if gather_mode is None:
            gather_mode = "torch"
        if gather_mode in ("rccl", "torch") and share and world > 1:
            sys.exit("bench.py: RCCL cannot run two ranks on one device (SI_BENCH_SHARE_DEVICE needs --gather p2p)")

    mg = si.modelgen
    with tempfile.TemporaryDirectory(prefix="si_bench_r%d_" % rank) as td:
        # the CPU legs run FIRST (bounded, ~20 s): the GPU leg is then the long, uninterrupted tail of the run
        cpu = {"cpu_baseline": None}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, mg, td)

        builder, shape = build_model(mg, args.model, args.batch, args.size)
        pp, bp = os.path.join(td, "m.pnnx.param"), os.path.join(td, "m.pnnx.bin")
        builder.save(pp, bp)
        flops_step = mg.conv_flops(builder)

        extra_opts = {kv.split("=", 1)[0]: int(kv.split("=", 1)[1]) for kv in args.engine_opt}
        e = si.Engine(device=dev, outputs_to_host=0, graph=args.graph, winograd=args.winograd, fp16=args.fp16, **extra_opts)
        try:
            e.load_model(pp, bp)
        except si.StatusError as ex:
            sys.exit("bench.py: %s cannot be loaded with these options (%s)%s" % (
                args.model, ex, "; the fp16 storage path has no kernels for some of its layers -- see the engine's log line above" if args.fp16 else ""))
        iname, oname = e.input_names()[0], e.output_names()[0]
        lanes = e.schedule().get("lanes", 1)   # 2: the batch runs as two half-batch lanes on two streams (engine option "streams")
        # global batch = per-GPU batch * world; this rank's slab gets its own seed (distinct images)
        x = mg.synth_input(shape, seed=1 + rank)
        dx = hipops.DeviceBuffer.from_numpy(x)
        e.input_device(iname, dx.ptr)
        oshape = e.operand_shape(oname)

        sf = og = None
        if gather_mode in ("p2p", "rccl"):
            try:
                # collective: every rank succeeds or none does.  auto: si_gather_create_mode(SI_GATHER_AUTO) itself falls back to
                # RCCL on every rank when the direct path cannot be set up on one
                sf = shard.ShardedForward(e, oname, group, dev, slots=4,
                                          mode={"p2p": "direct", "rccl": "rccl"}[gather_mode] if args.gather != "auto" else "auto")
                if sf.gather.mode == "rccl" and gather_mode == "p2p":
                    gather_note = "direct gather could not be set up on every rank; RCCL through the C-ABI instead"
                    gather_mode = "rccl"
            except shard.ShardError as ex:
                if args.gather != "auto":
                    raise
                gather_note = "neither the direct nor the RCCL gather of include/si_shard.h could be set up (%s)" % ex
                gather_mode = "torch"
                if share and world > 1:
                    sys.exit("bench.py: RCCL cannot run two ranks on one device")
        def step():
            if sf is not None:
                # Forward() writes this step's [B/G, rows, 85] slab straight into its place in one of three gathered buffers
                # (Engine::Output), the slab is fanned out to every peer behind it, and the PREVIOUS step's gather completes
                sf.forward()
            elif og is not None:
                e.bind_output(oname, og.target().data_ptr())
                e.forward()          # synchronous: kernels of this step are done when it returns
                og.submit_inplace()  # all-gather of THIS step's slab, overlapped with the next step's compute
            else:
                e.forward()

        def fence():
            if sf is not None:
                sf.flush()           # the last step's gather has completed on every rank (node barrier inside)
                group.barrier()
            elif og is not None:
                og.drain()           # every gather issued so far has completed
                dist.barrier()
                torch.cuda.synchronize()
            H.si_hip_device_sync()

        def max_over_ranks(v):
            if sf is not None:
                return group.max_f64(v)
            if og is not None:
                t = torch.tensor([v], dtype=torch.float64, device="cuda:%d" % dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item())
            return v

        def slab_checksums():
            """after a fence: does every rank's gathered buffer hold every rank's slab?  Compared through 64-bit sums of the
            raw words, exchanged host-side."""
            n = int(np.prod(oshape))
            if sf is not None:
                full = hipops.DeviceBuffer.view(sf.gathered_ptr(), n * 4 * world).to_numpy((world, n), np.uint32)
            else:
                full = og.latest().view(torch.int32).reshape(world, n).cpu().numpy().view(np.uint32)
            sums = full.astype(np.uint64).sum(axis=1)
            if sf is not None:
                theirs = np.frombuffer(b"".join(group.allgather_bytes(np.uint64(sums[rank]).tobytes())), np.uint64)
            else:
                t = torch.tensor([int(sums[rank] >> np.uint64(1))], dtype=torch.int64, device="cuda:%d" % dev)
                allt = [torch.zeros_like(t) for _ in range(world)]
                dist.all_gather(allt, t)
                theirs = np.array([int(v.item()) for v in allt], np.uint64)
                sums = sums >> np.uint64(1)
            if not np.array_equal(sums, theirs) or int(sums[rank]) == 0:
                raise RuntimeError("rank %d: the gathered buffer does not hold every rank's slab (%s vs %s)" % (rank, sums, theirs))

        def setup_rccl():
            nonlocal torch, dist, og
            import torch as _torch  # plumbing only: process group, barrier, the RCCL all-gather
            torch = _torch
            torch.cuda.set_device(dev)
            dist = sd.init_process_group("nccl", device_index=dev)
            e.bind_output(oname, None)
            e.forward()
            optr, _ = e.extract_ptr(oname)
            og = sd.OverlappedGather(sd.as_torch(optr, oshape, dev))

        if gather_mode == "torch":
            setup_rccl()

        # warm-up; with the direct gather in `auto` mode it doubles as the acceptance test of that path: every rank reports
        # whether its steps ran and its gathered buffer checks out, and unless ALL do, all fall back to RCCL together
        if sf is not None and args.gather == "auto" and sf.gather.mode == "direct":
            ok = 1
            try:
                for _ in range(max(args.warmup, 2)):
                    step()
                fence()
                slab_checksums()
                fence()
            except Exception as ex:  # noqa: BLE001 -- any failure of the optional path means "use the fallback"
                ok = 0
                gather_note = "direct gather failed its warm-up check (%s)" % ex
            # the vote goes through a FRESH rendezvous: a warm-up that failed with a timeout leaves `group` dead (a timed-out
            # barrier poisons it), and a vote read from its stale bytes would mean nothing
            try:
                vote = shard.NodeGroup(shard.default_group_name() + "_vote", rank, world, timeout_s=180.0)
                all_ok = min(b[0] for b in vote.allgather_bytes(bytes([ok])))
                vote.close()
            except shard.ShardError as ex:
                sys.exit("bench.py: rank %d: the ranks could not agree on the gather path (%s)" % (rank, ex))
            if not all_ok:
                if gather_note is None:
                    gather_note = "direct gather failed its warm-up check on another rank"
                try:
                    sf.close()
                except Exception:  # noqa: BLE001
                    pass
                sf = None
                if share and world > 1:
                    sys.exit("bench.py: RCCL cannot run two ranks on one device")
                try:
                    # (a fresh rendezvous: a timed-out barrier leaves the old group dead)
                    group = shard.NodeGroup(shard.default_group_name() + "_rccl", rank, world, timeout_s=180.0)
                    sf = shard.ShardedForward(e, oname, group, dev, slots=4, mode="rccl")
                    gather_mode = "rccl"
                except shard.ShardError as ex:
                    gather_note += "; RCCL through the C-ABI failed too (%s)" % ex
                    gather_mode = "torch"
                    setup_rccl()
        for _ in range(args.warmup):
            step()
        fence()
        if use_dist and args.warmup > 0:
            slab_checksums()
            fence()

        def timed_region():
            """K steps between two fences, repeated until --min-time of timed work has accumulated"""
            if sf is not None:
                sf.gather.stats(reset=True)
            windows, fwd_ms, total = [], 0.0, 0.0
            while True:
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                    fwd_ms += e.last_forward_ms()
                fence()
                dt = max_over_ranks(time.perf_counter() - t0) if use_dist else time.perf_counter() - t0
                windows.append(dt)
                total += dt
                if total >= args.min_time or len(windows) >= args.max_windows:
                    break
            gather_diag = None
            if sf is not None:
                # where a step's time went: host waits inside si_gather_complete (0 = the fan-out hid behind the next step's
                # compute) and the device time of the peer copies; every rank reports, rank 0 prints its own and the worst
                gs = sf.gather.stats()
                waits = group.allgather_f64(gs["gather_wait_ms"])
                copies = group.allgather_f64(gs["peer_copy_ms"] or 0.0)
                gather_diag = dict(gs)
                gather_diag.update({"transport": sf.gather.mode,
                                    "gather_wait_ms_max_over_ranks": round(float(waits.max()), 4),
                                    "peer_copy_ms_max_over_ranks": round(float(copies.max()), 4),
                                    "slab_mbytes": round(sf.gather.slab_bytes / 1e6, 2), "slots": sf.gather.slots,
                                    "copy_engine": ("ncclAllGather, in place, on the gather's own stream (si_rccl_allgather)" if sf.gather.mode == "rccl" else
                                                    "hipMemcpyDtoDAsync per peer on its own stream") +
                                                   "; HSA_ENABLE_SDMA=%s (unset / 1: SDMA engines, 0: blit kernels that take CUs from the "
                                                   "convs -- a rocprofv3 kernel trace shows them as __amd_rocclr_copyBuffer)" % os.environ.get("HSA_ENABLE_SDMA", "unset")})
            return windows, fwd_ms, total, gather_diag

        def median(ws):
            w = sorted(ws)
            return w[len(w) // 2] if len(w) % 2 else 0.5 * (w[len(w) // 2 - 1] + w[len(w) // 2])

        windows, fwd_ms, total, gather_diag = timed_region()
        gather_ab = None
        if use_dist and args.gather == "both" and sf is not None:
            # the same engine, the same run: the direct fan-out was timed above, now RCCL's all-gather through the C-ABI
            slab_checksums()
            fence()
            gather_ab = {"p2p": {"value": round(args.batch * world * args.steps / median(windows), 2),
                                 "ms_per_step": round(median(windows) / args.steps * 1e3, 3), "gather": gather_diag}}
            group.barrier()
            sf.close()
            try:
                sf = shard.ShardedForward(e, oname, group, dev, slots=4, mode="rccl")
                for _ in range(max(args.warmup, 2)):
                    step()
                fence()
                slab_checksums()
                fence()
                w2, _, _, gd2 = timed_region()
                gather_ab["rccl"] = {"value": round(args.batch * world * args.steps / median(w2), 2),
                                     "ms_per_step": round(median(w2) / args.steps * 1e3, 3), "gather": gd2}
            except shard.ShardError as ex:   # (collective: every rank fails the same way)
                gather_ab["rccl"] = {"error": str(ex)}
                sf = None
        if use_dist and (sf is not None or og is not None):
            slab_checksums()
        wsorted = sorted(windows)
        dt = median(windows)
        fwd_ms_per_step = fwd_ms / (args.steps * len(windows))

        roof, agg, layers = None, {}, None
        if rank == 0:
            if sf is not None:
                e.bind_output(oname, None)
            passes = [e.profile() for _ in range(max(args.profile_passes, 1))]
            layers = passes[-1]
            roof, agg = roofline_from_profile(passes, fp16=bool(args.fp16), workload=workload_key(args, shape))
            if args.layers:
                # per layer: achieved TF/s and GB/s, the roofline the layer's ALGORITHMIC work sits under (the longer of its FLOPs at
                # the matrix peak of the precision and its bytes at 8 TB/s) and the fraction of that bound it reaches
                peak_tf = PEAK_F16_MFMA_TFLOPS if args.fp16 else PEAK_FP32_MFMA_TFLOPS
                for L in layers:
                    if L["kernel"].startswith("aliased"):
                        # a concat whose producers wrote straight into its buffer: no launch -- the interval is the event pair itself
                        print("%-28s %-22s %-48s %8.3f ms   (the event pair; producers write their slices in place)" % (L["name"], L["type"], L["kernel"], L["ms"]),
                              file=sys.stderr)
                        continue
                    tf = L["flops"] / (L["ms"] * 1e-3) / 1e12 if L["ms"] > 0 else 0
                    gb = L["bytes"] / (L["ms"] * 1e-3) / 1e9 if L["ms"] > 0 else 0
                    t_f, t_b = L["flops"] / (peak_tf * 1e12), L["bytes"] / (PEAK_HBM_GBS * 1e9)
                    bound = "mfma" if t_f >= t_b else "hbm"
                    frac = max(t_f, t_b) / (L["ms"] * 1e-3) if L["ms"] > 0 else 0
                    print("%-28s %-22s %-48s %8.3f ms %7.1f TF/s %8.1f GB/s  %-4s %5.2f" % (L["name"], L["type"], L["kernel"], L["ms"], tf, gb, bound, frac),
                          file=sys.stderr)
                print("sum of layer times %.3f ms" % sum(L["ms"] for L in layers), file=sys.stderr)

        aux = {}
        if rank == 0 and world == 1 and not args.no_aux:
            aux = aux_measurements(args, si, hipops, H, e, oname, oshape, pp, bp, dev, x, extra_opts)
            if args.model == "yolov5s" and len(oshape) == 3 and not args.fp16:
                aux["app_pipeline"] = app_pipeline(args, si, hipops, H, e, iname, oname, oshape, args.batch * args.steps / dt)
                e.input_device(iname, dx.ptr)
        secondary = None
        if rank == 0 and world == 1 and not args.no_secondary and args.model == "yolov5s" and args.batch == 32 and args.size == 640 and not args.fp16:
            # the headline engine is done (nothing below reads it): released first.  Measured (profiles/r05_secondary_gap.txt): with it
            # still alive the fp16 / f32_split engines created behind it read 3 % / 1.5 % low (25.5 -> 24.7 k, 9.48 -> 9.33 k; the fp32 cases
            # do not move), in front of it or with it released they read what a process of their own reads
            e.release()
            secondary = secondary_measurements(args, si, hipops, H, mg, td, dev)

        if sf is not None:
            group.barrier()
            sf.close()
            group.close()
        elif og is not None:
            dist.barrier()

    if dist is not None:
        dist.destroy_process_group()
    if rank != 0:
        return
    imgs = args.batch * world * args.steps
    value = imgs / dt
    prec = "fp16" if args.fp16 else "fp32"
    ceiling = (PEAK_F16_MFMA_TFLOPS if args.fp16 else PEAK_FP32_MFMA_TFLOPS) * 1e12 / (flops_step / args.batch)  # images/s/GPU at the MFMA peak
    strong = bool(args.global_batch)
    gather_text = {"p2p": "direct fan-out over IPC-shared HBM (include/si_shard.h), one device-to-device copy per peer",
                   "rccl": "RCCL ncclAllGather through the C-ABI (si_rccl_allgather, include/si_shard.h; no torch)",
                   "torch": "RCCL through torch.distributed all_gather_into_tensor"}.get(gather_mode)
    out = {
        "metric": "images/sec %s %dx%d %s batch=%d per GPU, Engine::Forward()" % (
            {"yolov5s": "YOLOv5s", "resnet18": "ResNet18", "mobilenetv3": "MobileNetV3-Small"}[args.model], shape[1], shape[2], prec, args.batch),
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f16" if args.fp16 else "f32", "data": "synthetic",
        "config": {"workload": "%s %dx%d %s forward, batch %d per GPU (global %d), random-init weights, "
                               "inputs resident in HBM%s" % (args.model, shape[1], shape[2], prec, args.batch,
                                                            args.batch * world,
                                                            (", every step's output slab all-gathered (%s), overlapped with the next step" % gather_text) if use_dist else ""),
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                   "workload_key": workload_key(args, shape),
                   "gather": gather_mode, "gather_note": gather_note, "engine_options": extra_opts, "lanes": lanes,
                   "hipgraph": bool(args.graph), "winograd_for_3x3s1": {0: "off", 1: "F(2,3)", 2: "F(4,3)"}.get(args.winograd, "F(2,3)")},
        "windows": {"count": len(windows), "steps_each": args.steps, "timed_s_total": round(total, 3),
                    "value_is": "median window",
                    "ms_per_step_median": round(dt / args.steps * 1e3, 3),
                    "ms_per_step_min": round(wsorted[0] / args.steps * 1e3, 3),
                    "ms_per_step_max": round(wsorted[-1] / args.steps * 1e3, 3),
                    "ms_per_step_first": round(windows[0] / args.steps * 1e3, 3)},
        "forward_kernel_ms_per_step": round(fwd_ms_per_step, 3),
        "gather": gather_diag,
        "gather_ab": gather_ab,
        # what a step waited for: its own peer copies (the gather is the bound), the node barrier (the slowest rank: imbalance, or
        # ranks sharing a device), or neither
        "step_bound": (None if gather_diag is None else
                       ("gather (waiting for the peer copies)" if gather_diag["gather_wait_copies_ms"] > 0.1 * (dt / args.steps * 1e3) else
                        "slowest rank (waiting in the node barrier)" if gather_diag["gather_wait_barrier_ms"] > 0.1 * (dt / args.steps * 1e3) else
                        "compute")),
        "gflop_per_image": round(flops_step / args.batch / 1e9, 3),
        "frac_of_mfma_ceiling": round(value / world / ceiling, 4),
        "roofline": roof,
        "conv_kernels": {k: {"ms_per_step": round(v["ms"] / max(args.profile_passes, 1), 3),
                             "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else 0}
                         for k, v in agg.items()},
    }
    out.update(cpu)
    out.update(aux)
    if secondary is not None:
        out["secondary"] = secondary
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

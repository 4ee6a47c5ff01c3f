#!/usr/bin/env python3
"""Benchmark of the SingleHDR hot path on MI355X (driver contract: one JSON line on rank 0).

Workload (BASELINE.json configs[2], the configuration the headline metric
"HDR images/sec/GPU (512x512) end-to-end" is quoted on): full
deq -> clip -> lin -> apply_rf -> alpha -> hal -> blend inference, batch 16 of
512x512 synthetic LDR images per GPU, fp32, random-init weights (Keras
initialisers, non-trivial BatchNorm statistics).  One "step" = one pass of
the hot path over one batch, inputs already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver through torch.distributed.run (one rank per
GPU); inference shards over the batch axis with no data-path collective
("weak" scaling: every rank processes its own batch).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_{32x32x2,16x16x4}_f32 dense peak
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (the 5 PF headline figure includes 2:1 sparsity)
TRAFFIC_FILE = next((f for f in ("r03_traffic.json", "r02_traffic.json", "r01_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", f))), "r01_traffic.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-refinement", action="store_true", help="skip the extra inference leg with the Refinement-Net")
    ap.add_argument("--layers", action="store_true", help="print a per-conv-launch table to stderr")
    ap.add_argument("--streams", type=int, default=2, help="inference: batch slices run on this many HIP streams")
    ap.add_argument("--exact-fp32", action="store_true", help="every layer on the exact-fp32 kernels (no split-operand fp16 kernels)")
    ap.add_argument("--graph", action="store_true", help="replay the forward as one HIP graph (small-batch latency)")
    ap.add_argument("--train-steps", type=int, default=3, help="timed joint-training steps (0 = skip that leg)")
    ap.add_argument("--train-batch", type=int, default=32)
    ap.add_argument("--train-size", type=int, default=256)
    ap.add_argument("--single-stream", action="store_true", help="joint step: run deq / lin / hal on one stream")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals on one card)")
    ap.add_argument("--finetune-steps", type=int, default=2, help="timed fine-tuning steps per precision (0 = skip that leg)")
    ap.add_argument("--finetune-batch", type=int, default=4)       # finetune_real_dataset.py:25
    ap.add_argument("--finetune-size", type=int, default=1024)
    ap.add_argument("--finetune-prec", default="fp32,fp16", help="comma list of the fine-tuning precisions to time")
    return ap.parse_args()


def randomise_bn(model, gen):
    """BN moving stats mu~N(0,0.1), var~U(0.5,1.5); gamma/beta non-trivial."""
    with torch.no_grad():
        for name, t, _ in model.named_weights():
            if name.endswith(".moving_mean") or name.endswith(".beta"):
                t.copy_(torch.randn(t.shape, generator=gen) * 0.1)
            elif name.endswith(".moving_variance") or name.endswith(".gamma"):
                t.copy_(torch.rand(t.shape, generator=gen) + 0.5)
            elif name.endswith(".bias"):
                t.copy_(torch.randn(t.shape, generator=gen) * 0.05)


def conv_flops(x, w, stride, cout_valid=None):
    """Algorithmic FLOPs 2*Ho*Wo*Cin*Cout*kh*kw of the REFERENCE layer: zero-padded filter
    channels (Cin 3->4, 9->12, 93->96; Cout 3->16) are not counted."""
    n, h, wd, _ = x.shape
    kh, kw, cin, cout = w.shape
    cout = cout_valid or cout
    cin = {4: 3, 12: 9, 96: 93}.get(cin, cin)
    ho, wo = -(-h // stride), -(-wd // stride)
    return 2.0 * n * ho * wo * cin * cout * kh * kw


def conv_variant(w, x, x2, algo, stride=1):
    """Which kernel shdr_conv2d_fwd_f32 dispatches to (mirrors csrc/conv.hip)."""
    c1 = x.shape[3]
    c2 = 0 if x2 is None else x2.shape[3]
    cout = w.shape[3]
    mfma_ok = c1 % 4 == 0 and c2 % 4 == 0 and cout % 16 == 0
    if algo == 2 or (algo == 0 and not mfma_ok):
        return "conv_direct_kernel"
    narrow = (c2 == 0 and c1 in (4, 8, 12, 16)) or (c1 == 16 and c2 == 16)                 # rega_ok() of conv.hip
    if algo == 0 and stride == 1 and cout in (16, 32) and narrow and w.shape[0] * w.shape[1] * (c1 + c2) * cout * 4 <= 100 * 1024:
        return "conv_rega_kernel<%d>" % cout
    if algo == 0 and stride == 1 and cout == 64 and c1 == 4 and c2 == 0 and tuple(w.shape[:2]) == (3, 3):
        return "conv_rega_kernel<64>"
    k = w.shape[0] * w.shape[1] * (c1 + c2)
    bn = 128 if (cout % 128 == 0 and k > 128) else 64 if cout % 64 == 0 else 32 if cout % 32 == 0 else 16
    return "conv_mfma_dma_kernel<128,%d>" % bn


def host_cores():
    """CPU threads this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands a container
    the affinity of the whole host but the quota of its share -- hundreds of OpenMP threads spinning on 16 CPUs' worth of time
    take minutes per convolution)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, q // int(f2.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 32))


def progress(msg):
    """one line on stderr per leg: the one JSON line on stdout stays alone, and a long run is never silent"""
    print("[bench] " + msg, file=sys.stderr, flush=True)


def fp16_roofline(K, fstep, ldr, hdr):
    """Per-launch HIP-event timing of the fp16 conv kernels inside ONE extra fine-tuning step (on the launch stream): the
    dominant kernel family by time, its executed = algorithmic FLOPs (no Winograd here) against the dense fp16 MFMA peak."""
    import ctypes
    lib = K._lib.load()
    recs = []
    orig_c, orig_w = K.conv2d_h, K.conv2d_wgrad_h

    def conv_h(x, wp, bias, khw, cout_gemm, stride=1, x2=None, act1=0, cout_valid=None, pad=None, out_hw=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig_c(x, wp, bias, khw, cout_gemm, stride=stride, x2=x2, act1=act1, cout_valid=cout_valid, pad=pad, out_hw=out_hw)
        e1.record()
        c2 = 0 if x2 is None else x2.shape[3]
        d = K._conv_desc_h(x.shape, c2, khw, cout_gemm, stride, None, pad, out_hw)
        # the dispatch order of csrc/conv_f16.hip (shdr_conv2d_fwd_f16), asked from the library's own predicates
        if lib.shdr_conv2d_patch_ok_f16(ctypes.byref(d)):
            label = "conv_f16_patch_kernel"
        elif y.dtype == torch.float16 and lib.shdr_conv2d_w3_ok_f16(ctypes.byref(d)):
            label = "conv_f16_w3_kernel"
        else:
            label = "conv_f16_kernel<%s>" % ("128,128" if cout_gemm % 128 == 0 else "256,64" if cout_gemm % 64 == 0 else
                                              "256,32" if cout_gemm % 32 == 0 else "256,16")
        recs.append((label, 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * (x.shape[3] + c2) * cout_gemm * khw[0] * khw[1], e0, e1))
        return y

    def wgrad_h(x, x2, dz, w_shape, stride=1, x2_scale=1.0, cout_valid=None, out=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dw = orig_w(x, x2, dz, w_shape, stride, x2_scale, cout_valid=cout_valid, out=out)
        e1.record()
        c2 = 0 if x2 is None else x2.shape[3]
        d = K._conv_desc_h(x.shape, c2, tuple(w_shape[:2]), w_shape[3], stride, cout_valid)
        label = "wgrad_f16_alltaps_kernel" if lib.shdr_conv2d_wgrad_alltaps_ok_f16(ctypes.byref(d), 0, dz.shape[3]) else "wgrad_f16_kernel"
        recs.append((label, 2.0 * dz.shape[0] * dz.shape[1] * dz.shape[2] * w_shape[2] * w_shape[3] * w_shape[0] * w_shape[1], e0, e1))
        return dw

    K.conv2d_h, K.conv2d_wgrad_h = conv_h, wgrad_h
    try:
        fstep(ldr, hdr, apply=False)
        torch.cuda.synchronize()
    finally:
        K.conv2d_h, K.conv2d_wgrad_h = orig_c, orig_w
    agg = {}
    for label, fl, e0, e1 in recs:
        a = agg.setdefault(label, [0.0, 0.0, 0])
        a[0] += fl; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
    dom = max(agg, key=lambda k: agg[k][1])
    fl, sec, cnt = agg[dom]
    tot_fl, tot_sec = sum(a[0] for a in agg.values()), sum(a[1] for a in agg.values())
    return {"bound": "mfma", "kernel": dom, "achieved": round(fl / sec / 1e12, 1), "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(fl / sec / 1e12 / F16_MFMA_PEAK_TFLOPS, 4), "launches_per_step": cnt, "avg_launch_ms": round(sec / cnt * 1e3, 4),
            "algorithmic_gflop_per_launch": round(fl / cnt / 1e9, 2), "traffic": None,
            "all_conv": {"tflops": round(tot_fl / tot_sec / 1e12, 1), "ms_per_step": round(tot_sec * 1e3, 2),
                         "gflop_per_step": round(tot_fl / 1e9, 1)},
            "per_kernel": {k: {"tflops": round(v[0] / v[1] / 1e12, 1), "ms_per_step": round(v[1] * 1e3, 2), "launches_per_step": v[2]}
                           for k, v in sorted(agg.items())}}


def train_roofline(K, run_step, layers=False):
    """Per-launch HIP-event timing (on the launch streams) of every conv forward / input-gradient / weight-gradient call inside ONE
    extra training step: the dominant kernel family by time with the MFMA FLOPs it executes against the peak of its matrix pipe.
    Labels are the library's plans (asked, not guessed): forward and dgrad by shdr_conv2d_plan_f32 of the (transposed) layer, the
    weight gradient by the rule of _ops.conv2d_wgrad (Winograd-domain kernel or the MFMA / all-taps family)."""
    recs = []
    orig_c, orig_d, orig_w = K.conv2d, K.conv2d_dgrad, K.conv2d_wgrad
    depth = [0]

    def ev():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    plan_kernel = {"x3": "conv_x3_kernel", "x3n": "conv_x3n_kernel", "fused": "winograd_fused_kernel", "planes": "winograd_planes",
                   "mfma": "conv_mfma_dma_kernel", "direct": "conv_direct_kernel"}

    def conv(x, w, bias=None, stride=1, x2=None, **kw):
        if depth[0] or x.dtype != torch.float32:
            return orig_c(x, w, bias, stride=stride, x2=x2, **kw)
        c2 = 0 if x2 is None else x2.shape[3]
        plan = K.conv2d_plan(tuple(x.shape), tuple(w.shape), c2=c2, stride=stride, x2_scale=kw.get("x2_scale", 1.0),
                             has_residual=kw.get("residual") is not None, cout_valid=kw.get("cout_valid"))
        e0, e1 = ev()
        e0.record()
        depth[0] += 1
        try:
            y = orig_c(x, w, bias, stride=stride, x2=x2, **kw)
        finally:
            depth[0] -= 1
        e1.record()
        recs.append(("fwd " + plan_kernel[plan], conv_flops(x, w, stride, kw.get("cout_valid")), e0, e1,
                     "%dx%dx%d %d+%d->%d k%d s%d" % (x.shape[0], x.shape[1], x.shape[2], x.shape[3], c2, w.shape[3], w.shape[0], stride)))
        return y

    def dgrad(dz, w, x_shape, c1, c2, which, stride=1, x2_scale=1.0):
        kh, kw_, _, cout = w.shape
        plan = K.conv2d_plan((x_shape[0], dz.shape[1], dz.shape[2], dz.shape[3]), (kh, kw_, dz.shape[3], x_shape[3])) if stride == 1 else "mfma"
        e0, e1 = ev()
        e0.record()
        depth[0] += 1
        try:
            dx = orig_d(dz, w, x_shape, c1, c2, which, stride, x2_scale)
        finally:
            depth[0] -= 1
        e1.record()
        recs.append(("dgrad " + plan_kernel[plan], 2.0 * dz.shape[0] * dz.shape[1] * dz.shape[2] * x_shape[3] * dz.shape[3] * kh * kw_, e0, e1,
                     "%dx%dx%d %d<-%d k%d s%d" % (x_shape[0], x_shape[1], x_shape[2], x_shape[3], dz.shape[3], kh, stride)))
        return dx

    def wgrad(x, x2, dz, w_shape, stride=1, x2_scale=1.0, out=None):
        kh, kw_, cin, cout = w_shape
        c1, c2 = x.shape[3], (0 if x2 is None else x2.shape[3])
        wino = (K.WINOGRAD and (kh, kw_) == (3, 3) and stride == 1 and c1 % 32 == 0 and c2 % 32 == 0 and cout % 64 == 0 and c1 % 16 == 0)
        e0, e1 = ev()
        e0.record()
        depth[0] += 1
        try:
            dw = orig_w(x, x2, dz, w_shape, stride, x2_scale, out=out)
        finally:
            depth[0] -= 1
        e1.record()
        if depth[0] == 0:
            recs.append(("wgrad_winograd_kernel" if wino else "wgrad_mfma / wgrad_alltaps_kernel",
                         2.0 * dz.shape[0] * dz.shape[1] * dz.shape[2] * min(cin, {4: 3, 12: 9, 96: 93}.get(cin, cin)) * cout * kh * kw_, e0, e1,
                         "%dx%dx%d %d+%d->%d k%d s%d" % (x.shape[0], x.shape[1], x.shape[2], c1, c2, cout, kh, stride)))
        return dw

    K.conv2d, K.conv2d_dgrad, K.conv2d_wgrad = conv, dgrad, wgrad
    try:
        run_step()
        torch.cuda.synchronize()
    finally:
        K.conv2d, K.conv2d_dgrad, K.conv2d_wgrad = orig_c, orig_d, orig_w
    agg = {}
    for label, fl, e0, e1, desc in recs:
        a = agg.setdefault(label, [0.0, 0.0, 0])
        a[0] += fl; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
        if layers:
            print("%-36s %-34s %8.3f ms %7.1f TF alg" % (label, desc, e0.elapsed_time(e1), fl / e0.elapsed_time(e1) / 1e9), file=sys.stderr)

    def executed(label, fl):          # MFMA FLOPs the kernel executes, and the peak of the pipe it runs on
        if "x3" in label:
            return 3.0 * fl, F16_MFMA_PEAK_TFLOPS
        if "winograd" in label:
            return fl / 2.25, F32_MFMA_PEAK_TFLOPS
        return fl, F32_MFMA_PEAK_TFLOPS
    dom = max(agg, key=lambda k: agg[k][1])
    fl, sec, cnt = agg[dom]
    ex, peak = executed(dom, fl)
    tot_sec = sum(a[1] for a in agg.values())
    return {"bound": "mfma", "kernel": dom, "achieved": round(ex / sec / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
            "frac": round(ex / sec / 1e12 / peak, 4), "launches_per_step": cnt, "avg_launch_ms": round(sec / cnt * 1e3, 4),
            "algorithmic_gflop_per_launch": round(fl / cnt / 1e9, 2), "executed_gflop_per_launch": round(ex / cnt / 1e9, 2),
            "traffic": None, "conv_ms_per_step": round(tot_sec * 1e3, 2),
            "note": "HIP events around each conv forward / dgrad / wgrad call of one extra step (calls on three streams overlap: the "
                    "per-family times add up to more than the step); executed FLOPs: split-operand kernels x 3 (fp16 pipe), Winograd "
                    "kernels / 2.25 (fp32 pipe)",
            "per_kernel": {k: {"executed_tflops": round(executed(k, v[0])[0] / v[1] / 1e12, 1), "frac_of_pipe_peak": round(executed(k, v[0])[0] / v[1] / 1e12 / executed(k, v[0])[1], 4),
                               "ms_per_step": round(v[1] * 1e3, 2), "launches_per_step": v[2]} for k, v in sorted(agg.items())}}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback for the hot path)"
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev                 # one rank per GPU; (rehearsals with --backend gloo may share a card)
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":                # RCCL over xGMI: the production path
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    pkg = importlib.import_module("singlehdr-tf2_amd")
    K = pkg._ops
    K.EXACT_FP32 = bool(args.exact_fp32)
    torch.manual_seed(1234)
    gen = torch.Generator().manual_seed(4321)
    deq = pkg.dequantization_net.model()
    lin = pkg.linearization_net.model()
    hal = pkg.hallucination_net.model()
    for m in (deq, lin, hal):
        randomise_bn(m, gen)
    run = (pkg.pipeline.GraphedInference(deq, lin, hal, None) if args.graph
           else pkg.pipeline.Inference(deq, lin, hal, None, streams=args.streams))
    eager = pkg.pipeline.Inference(deq, lin, hal, None)

    g = torch.Generator().manual_seed(3 + rank)
    ldr = (torch.round(torch.rand((args.batch, args.size, args.size, 3), generator=g) * 255.0) / 255.0).cuda()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = run(ldr)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run(ldr)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(out).all())
    if rank == 0:
        progress("inference leg done: %.2f ms/step" % (dt / args.steps * 1e3))
    n_gpus = world
    ms_per_step = dt / args.steps * 1e3
    value = args.batch * n_gpus * args.steps / dt

    result = {
        "metric": "HDR images/sec (512x512) end-to-end inference, deq+lin+hal",
        "value": round(value, 3), "unit": "images/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "dtype_note": "fp32 tensors, fp32 results; the wide 3x3 layers, the 7x7/2 stem, the 1x1 layers with K >= 256 (csrc/conv_x3.hip) and the "
                      "narrow U-Net layers (csrc/conv_x3n.hip) compute every fp32 product as three fp16 MFMA products of split operands with "
                      "fp32 accumulation (3 * 2^-22 per product, held to the exact-fp32 kernels' 1e-5 bar against the float64 oracle).  RANGE: these "
                      "kernels scale their input by the power of two that brings its range slot (max |x|, written by the producing kernel's "
                      "epilogue or measured below the C ABI) to [2^10, 2^11) and undo it in the epilogue, both exact: any finite fp32 input "
                      "range is taken, elements down to 2^-25 of the tensor maximum keep 22 mantissa bits, non-finite inputs give "
                      "non-finite outputs on their receptive field as the exact kernels do (tests/test_gpu_ops.py::"
                      "test_split_operand_forward_is_range_safe: 1e5, 1e-7, 1e30, 3e-30, a 7e4 outlier, +-inf).  All other layers exact "
                      "fp32 (fp32 MFMA / FMA); --exact-fp32 runs every layer on the exact kernels",
        "config": {"workload": "BASELINE configs[2]: full deq+lin+hal inference, batch=%d x %dx%d per GPU, "
                               "fp32 tensors and results, histogram B=4/8/16" % (args.batch, args.size, args.size),
                   "per_gpu_batch": args.batch, "hip_streams_per_gpu": args.streams,
                   "parallelism": "batch-sharded x%d, no collective" % n_gpus},
    }
    if rank == 0 and n_gpus == 1 and not args.exact_fp32 and not args.no_roofline:      # (--no-roofline = the lean runs under the profiler)
        # the same leg with every layer on the exact-fp32 kernels (fp32 MFMA / FMA products only), and how far the two results are apart
        K.EXACT_FP32 = True
        try:
            for _ in range(2):
                out_x = run(ldr)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nx = max(3, args.steps // 4)
            for _ in range(nx):
                out_x = run(ldr)
            torch.cuda.synchronize()
            xdt = (time.perf_counter() - t0) / nx
            result["exact_fp32_kernels_only"] = {"ms_per_step": round(xdt * 1e3, 3), "images_per_s": round(args.batch / xdt, 3),
                                                 "max_abs_diff_over_max": float((out - out_x).abs().max() / out_x.abs().max())}
            del out_x
        finally:
            K.EXACT_FP32 = False

    # ---- BASELINE configs[1]: Dequantization-Net only, batch 8 x 512 x 512 (informational; the headline stays configs[2]) ----
    if not args.no_refinement:
        x8 = ldr[:8].contiguous() if args.batch >= 8 else ldr
        with torch.no_grad():
            for _ in range(2):
                deq(x8, training=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                deq(x8, training=False)
            torch.cuda.synchronize()
        ddt = (time.perf_counter() - t0) / args.steps
        result["deq_only"] = {"workload": "BASELINE configs[1]: Dequantization-Net forward, batch=%d x %dx%d, one GPU"
                                          % (x8.shape[0], args.size, args.size),
                              "ms_per_step": round(ddt * 1e3, 3), "images_per_s": round(x8.shape[0] / ddt, 2),
                              "tflops_algorithmic": round(37.83 * (args.size / 512.0) ** 2 * x8.shape[0] / ddt / 1e3, 2)}

    # ---- the HBM-bound kernels of the path against the HBM roofline (soft histogram standalone at B = 32,
    #      and the fused front end of the Linearization-Net as it runs inside the step) ----------
    if rank == 0 and not args.no_roofline:
        try:
            def hbm_rate(fn, nbytes, reps=10):
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                return {"ms": round(ms, 4), "algorithmic_bytes": int(nbytes), "achieved": round(nbytes / ms / 1e6, 1),
                        "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / ms / 1e6 / 8000.0, 4)}
            npix = args.batch * args.size * args.size
            result["hbm_kernels"] = {
                "soft_hist_kernel B=32 (standalone, not a reference value of B)":
                    hbm_rate(lambda: K.soft_hist(ldr, 32), npix * (12 + 384)),
                "lin_frontend_rows_kernel (image + sobel + B=4/8/16 histograms -> 96 ch)":
                    hbm_rate(lambda: K.lin_frontend(ldr, 96), npix * (12 + 384)),
            }
        except Exception as exc:      # an auxiliary leg must never cost the headline line
            result["hbm_kernels_error"] = repr(exc)[:300]

    # ---- the same inference with the Refinement-Net appended ("report with and without") ----------
    if not args.no_refinement:
        torch.manual_seed(4321)
        ref_net = pkg.refinement_net.model()
        run_ref = pkg.pipeline.Inference(deq, lin, hal, ref_net, streams=args.streams)
        for _ in range(2):
            out_ref = run_ref(ldr)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out_ref = run_ref(ldr)
        barrier()
        rdt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([rdt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rdt = float(t.item())
        assert bool(torch.isfinite(out_ref).all())
        result["with_refinement_net"] = {"workload": "deq+lin+hal+ref (test_real_refinement.py:86-110), same batch",
                                         "ms_per_step": round(rdt / args.steps * 1e3, 3),
                                         "images_per_s": round(args.batch * n_gpus * args.steps / rdt, 3)}
        del out_ref, run_ref, ref_net

    # ---- roofline of the dominant kernel: per-launch HIP-event timing on the launch stream ----------
    if rank == 0 and not args.no_roofline:
        progress("roofline leg (per-launch HIP events)")
        try:
            records = []     # (kernel label, flops, e0, e1, description, nested)
            orig = K.conv2d
            depth = [0]

            def x3_name(k, up=False):
                # the instantiation's name as rocprofv3 prints it (profiles/*_kernel_stats.csv): <UP, KH, KW, MP>; the 7x7 / 2 stem is the
                # phase-loop kernel on the 4 x 4 geometry
                if k == 7:
                    return "conv_x3_kernel<false, 4, 4, true>"
                return "conv_x3_kernel<%s, %d, %d, false>" % ("true" if up else "false", k, k)

            def timed_conv(x, w, bias=None, stride=1, x2=None, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                wino = None
                if (depth[0] == 0 and K.WINOGRAD and kw.get("algo", 0) == 0 and kw.get("pad") is None and kw.get("out_hw") is None
                        and not kw.get("w_batch_stride")):
                    # the kernel family the library plans for this layer (asked, not guessed): "x3", "fused", "planes", "mfma", "direct"
                    wino = K.conv2d_plan(tuple(x.shape), tuple(w.shape), c2=0 if x2 is None else x2.shape[3], stride=stride,
                                         x2_scale=kw.get("x2_scale", 1.0), has_residual=kw.get("residual") is not None,
                                         cout_valid=kw.get("cout_valid"))
                    if wino not in ("x3", "x3n", "fused", "planes"):
                        wino = None
                e0.record()
                depth[0] += 1
                try:
                    y = orig(x, w, bias, stride=stride, x2=x2, **kw)
                finally:
                    depth[0] -= 1
                e1.record()
                label = ("winograd_f2x2_3x3 (transforms + batched GEMM)" if wino == "planes" else
                         "winograd_fused_kernel" if wino == "fused" else x3_name(w.shape[0]) if wino == "x3" else
                         "conv_x3n_kernel" if wino == "x3n" else conv_variant(w, x, x2, kw.get("algo", 0), stride))
                records.append((label, conv_flops(x, w, stride, kw.get("cout_valid")), e0, e1,
                                "%dx%d %d+%d->%d k%d s%d" % (x.shape[1], x.shape[2], x.shape[3],
                                                            0 if x2 is None else x2.shape[3], w.shape[3], w.shape[0], stride),
                                depth[0] > 0))
                return y

            orig_cp = K.conv2d_maxpool2

            def timed_conv_pool(x, w, bias=None, act1=0, keep_y=True, proj=None):
                # conv + MaxPool2D(2) pairs of the Hallucination-Net encoder: one launch of the fused Winograd kernel where
                # it applies (timed here as that launch), otherwise conv2d() [recorded by timed_conv] + maxpool2
                plan = K.conv2d_plan(tuple(x.shape), tuple(w.shape)) if K.WINOGRAD and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 else None
                if plan not in ("fused", "x3"):            # direct plans: conv2d() [recorded by timed_conv] + maxpool2
                    if proj is not None:
                        return None                        # (the caller then runs the convolution and the 1x1 map as two calls)
                    y = timed_conv(x, w, bias, act1=act1)
                    return (y, K.maxpool2(y)) if keep_y else K.maxpool2(y)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = orig_cp(x, w, bias, act1, keep_y, proj)
                e1.record()
                if out is None:                            # projection refused by the library: nothing was launched
                    return None
                records.append(("winograd_fused_kernel" if plan == "fused" else x3_name(3), conv_flops(x, w, 1, None), e0, e1,
                                "%dx%d %d+0->%d k3 s1 +pool%s" % (x.shape[1], x.shape[2], x.shape[3], w.shape[3], " +proj" if proj is not None else ""), False))
                return out

            orig_ap = K.conv2d_avgpool2

            def timed_conv_avgpool(x, w, bias=None, act1=0, x2=None):
                # conv + AveragePooling2D(2) pairs of the U-Net encoders: the pooled tensor comes out of the split-operand kernels'
                # own epilogue (timed here as that launch), otherwise conv2d() [recorded by timed_conv] + avgpool2
                c2 = 0 if x2 is None else x2.shape[3]
                plan = K.conv2d_plan(tuple(x.shape), tuple(w.shape), c2=c2) if K.WINOGRAD and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 else None
                if plan not in ("x3", "x3n"):
                    y = timed_conv(x, w, bias, x2=x2, act1=act1)
                    return y, K.avgpool2(y)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = orig_ap(x, w, bias, act1, x2)
                e1.record()
                records.append((x3_name(w.shape[0]) if plan == "x3" else "conv_x3n_kernel", conv_flops(x, w, 1, None), e0, e1,
                                "%dx%d %d+%d->%d k%d s1 +avgpool" % (x.shape[1], x.shape[2], x.shape[3], c2, w.shape[3], w.shape[0]), False))
                return out

            orig_up = K.conv2d_up2

            def timed_conv_up2(x, w, bias=None, **kw):
                # bilinear 2x + conv of the decoders' `up` blocks: ONE launch of the fused Winograd kernel (up-sampling variant) where
                # the plan is "fused" -- timed here as that launch, with the conv layer's FLOPs --, otherwise resize2x + conv2d()
                # [the latter recorded by timed_conv]
                n, h, wd, c = x.shape
                plan = K.conv2d_plan((n, 2 * h, 2 * wd, c), tuple(w.shape)) if K.WINOGRAD else None
                if plan not in ("fused", "x3"):
                    if kw.get("proj") is not None:
                        return None
                    return timed_conv(K.resize2x(x), w, bias, **kw)      # direct plans: the up-sampled tensor goes through memory
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = orig_up(x, w, bias, **kw)
                e1.record()
                if out is None:
                    return None
                records.append(("winograd_fused_kernel" if plan == "fused" else x3_name(3, True), 2.0 * n * 4 * h * wd * c * w.shape[3] * 9, e0, e1,
                                "%dx%d(x2) %d+0->%d k3 s1 +bilinear" % (h, wd, c, w.shape[3]), False))
                return out

            reps = 3
            eager(ldr)           # the caching allocator's pool of THIS stream (the timed leg may have run on side streams)
            torch.cuda.synchronize()
            K.conv2d, K.conv2d_maxpool2, K.conv2d_up2, K.conv2d_avgpool2 = timed_conv, timed_conv_pool, timed_conv_up2, timed_conv_avgpool
            try:
                for _ in range(reps):
                    eager(ldr)
                torch.cuda.synchronize()
            finally:
                K.conv2d, K.conv2d_maxpool2, K.conv2d_up2, K.conv2d_avgpool2 = orig, orig_cp, orig_up, orig_ap
            # one entry per conv call of ONE pass, timed as the median over the `reps` passes (a host-side hiccup --
            # e.g. the runtime growing its signal pool inside hipEventRecord -- shows up as GPU idle time between
            # the two events of whichever call it hits, in one pass only)
            per_pass = len(records) // reps
            calls = []
            for i in range(per_pass):
                var, fl, _, _, desc, nested = records[i]
                ms = sorted(records[i + r * per_pass][2].elapsed_time(records[i + r * per_pass][3]) for r in range(reps))[reps // 2]
                calls.append((var, fl, ms * 1e-3, desc, nested))
            if args.layers:
                for var, fl, sec, desc, nested in calls:
                    print("%-46s %-28s %8.3f ms %7.2f TF%s" % (var, desc, sec * 1e3, fl / sec / 1e12,
                                                             "  (nested GEMM, executed FLOPs)" if nested else ""), file=sys.stderr)
            # layers: top-level calls with the reference layer's algorithmic FLOPs; kernels: every launch of a conv
            # kernel (top-level direct convs + the GEMMs nested in Winograd layers, with the FLOPs they execute)
            layers, agg = {}, {}
            is_x3 = lambda k: k.startswith("conv_x3")      # noqa: E731  (conv_x3_kernel<...> instantiations and conv_x3n_kernel)
            for var, fl, sec, desc, nested in calls:
                if not nested:
                    a = layers.setdefault(var, [0.0, 0.0, 0])
                    a[0] += fl; a[1] += sec; a[2] += 1
                if nested or not var.startswith("winograd_f2x2"):
                    a = agg.setdefault(var, [0.0, 0.0, 0])
                    # per KERNEL the MFMA FLOPs it executes: the fused Winograd kernel runs 16 / 36 of the layer's direct-form FLOPs,
                    # the split-operand kernel three fp16 MFMA products per fp32 product
                    # (a 32-cout layer on conv_x3 runs a 64-cout slice: twice its own FLOPs)
                    half = 2.0 if var.startswith("conv_x3_kernel") and "->32 k" in desc else 1.0
                    a[0] += fl / 2.25 if var == "winograd_fused_kernel" else 3.0 * half * fl if is_x3(var) else fl
                    a[1] += sec; a[2] += 1
            reps = 1             # `calls` holds one pass
            dom = max(agg, key=lambda k: agg[k][1])
            fl, sec, cnt = agg[dom]
            achieved = fl / sec / 1e12
            conv_total_flops = sum(a[0] for a in layers.values()) / reps
            conv_total_sec = sum(a[1] for a in layers.values()) / reps
            # HBM bytes per launch of the dominant kernel: PMC counters collected offline (rocprofv3 cannot profile
            # the process it runs in) with the same workload -- see profiles/r01_traffic.json for the recipe
            traffic = None
            try:
                with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)) as f:
                    tk = json.load(f)["kernels"]
                if dom.startswith("conv_mfma"):
                    key = dom.replace(",", ", ")[:-1]          # "conv_mfma_dma_kernel<128, 128"
                    hits = [v["hbm_bytes_per_launch"] for k, v in tk.items() if k.startswith(key) and k.endswith("true>")]
                    hits = [(h, 1) for h in hits[:1]]
                else:                                          # all template variants of the kernel, weighted by their launches
                    hits = [(v["hbm_bytes_per_launch"], v["launches"]) for k, v in tk.items() if k.startswith(dom)]
                traffic = int(sum(b * n for b, n in hits) / sum(n for _, n in hits)) if hits else None
            except (OSError, KeyError, ValueError):
                pass
            peak_of = lambda k: F16_MFMA_PEAK_TFLOPS if is_x3(k) else F32_MFMA_PEAK_TFLOPS      # noqa: E731
            to_alg = 2.25 if dom == "winograd_fused_kernel" else 1.0 / 3.0 if is_x3(dom) else 1.0
            # time the matrix pipes need for one step at their peaks (fp32-MFMA kernels against 157.3, the fp16 split kernel against 2500)
            floor_ms = sum(a[0] / (peak_of(k) * 1e12) for k, a in agg.items()) * 1e3
            result["roofline"] = {
                "bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": peak_of(dom),
                "unit": "TFLOP/s", "frac": round(achieved / peak_of(dom), 4),
                "mfma_dtype": "f16 (v_mfma_f32_16x16x32_f16, three products per fp32 product)" if is_x3(dom) else "f32 (v_mfma_f32_16x16x4_f32)",
                # SURVEY.md section 8(d) convention: the reference layers' direct-form fp32 FLOPs against the fp32-MFMA peak, no discount
                # for what Winograd saves and no surcharge for the three fp16 products of the split kernel
                "frac_algorithmic": round(achieved * to_alg / F32_MFMA_PEAK_TFLOPS, 4),
                "whole_step": {"executed_gflop_f32_mfma": round(sum(a[0] for k, a in agg.items() if not is_x3(k)) / 1e9, 1),
                               "executed_gflop_f16_mfma": round(sum(a[0] for k, a in agg.items() if is_x3(k)) / 1e9, 1),
                               "mfma_floor_ms": round(floor_ms, 3),
                               "frac_executed": round(floor_ms / ms_per_step, 4),
                               "frac_algorithmic": round(conv_total_flops / (ms_per_step * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                               "note": "frac_executed = time the matrix pipes need for the step's executed MFMA FLOPs at their peaks / the step "
                                       "time of the headline leg (all kernels, 2 HIP streams); frac_algorithmic = direct-form fp32 conv FLOPs "
                                       "per second / the fp32-MFMA peak"},
                "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/%s)" % TRAFFIC_FILE,
                "launches_per_step": cnt // reps, "avg_launch_ms": round(sec / cnt * 1e3, 4),
                "algorithmic_gflop_per_launch": round(fl * to_alg / cnt / 1e9, 3),
                "executed_gflop_per_launch": round(fl / cnt / 1e9, 3),
                "flop_convention": "MFMA FLOPs the kernel executes (Winograd kernels: layer FLOPs / 2.25; split-operand kernel: layer FLOPs x 3, "
                                   "fp16); per_layer_path holds the reference layers' direct-form (algorithmic) FLOPs",
                "all_conv": {"tflops": round(conv_total_flops / conv_total_sec / 1e12, 2),
                             "ms_per_step": round(conv_total_sec * 1e3, 3),
                             "gflop_per_step": round(conv_total_flops / 1e9, 1)},
                "per_kernel": {k: {"tflops": round(v[0] / v[1] / 1e12, 2), "ms_per_step": round(v[1] / reps * 1e3, 3),
                                   "launches_per_step": v[2] // reps} for k, v in sorted(agg.items())},
                "per_layer_path": {k: {"algorithmic_tflops": round(v[0] / v[1] / 1e12, 2), "ms_per_step": round(v[1] / reps * 1e3, 3),
                                       "layers_per_step": v[2] // reps} for k, v in sorted(layers.items())},
            }
        except Exception as exc:      # an auxiliary leg must never cost the headline line
            result['roofline_error'] = repr(exc)[:300]


    # ---- CPU baseline: the fp32 torch-CPU channels-last restatement of the same path ("port": TF2-CPU is
    #      not installable here), all host cores, 2 warm-ups, median of 5 runs, on a bounded sample of the workload (one image of
    #      the batch); the NumPy float32 oracle checks the GPU result of that image.  Rank 0, N = 1 only. ------------------------
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        try:
            from oracle import nets, torch_cpu  # checker / baseline only -- never on the product path
            table = np.load(os.path.join(ROOT, "singlehdr-tf2_amd", "data", "invemor_g0_hinv11.npy"))
            params = {k: {n: t.cpu().numpy() for n, t in m.state_dict().items()}
                      for k, m in (("deq", deq), ("lin", lin), ("hal", hal))}
            sample = ldr[:1].cpu().numpy()
            cores = host_cores()
            torch.set_num_threads(cores)
            progress("cpu_baseline: torch-CPU restatement on %d threads" % cores)
            cpu_model = "unknown"
            try:
                with open("/proc/cpuinfo") as f:
                    cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
            except OSError:
                pass
            cnets = {k: torch_cpu.Net(v) for k, v in params.items()}
            times = []
            for i in range(7):
                t0 = time.perf_counter()
                cpu_out = torch_cpu.inference(cnets, sample, table)
                if i >= 2:
                    times.append(time.perf_counter() - t0)
            times.sort()
            cpu_dt = times[len(times) // 2]
            gpu_img = out[:1].cpu().numpy()
            progress("cpu_baseline: %.2f s per image; NumPy oracle check of the GPU result" % cpu_dt)
            t0 = time.perf_counter()
            ref = nets.inference(params, sample, table, with_refinement=False)["A_pred"]
            np_dt = time.perf_counter() - t0
            result["cpu_baseline"] = {
                "value": round(1.0 / cpu_dt, 4), "unit": "images/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
                "sample": "1 image %dx%d of the batch, deq+lin+hal, fp32 torch-CPU channels-last restatement (oracle/torch_cpu.py: "
                          "proxy for TF2-CPU, which is not installable here); 2 warm-ups, median of 5 runs = %.2f s per image"
                          % (args.size, args.size, cpu_dt),
                "runs_s": [round(t, 3) for t in times],
                "cpu_vs_oracle_rel_err": float("%.3g" % (np.abs(cpu_out - ref).max() / np.abs(ref).max())),
                "gpu_vs_oracle_rel_err": float("%.3g" % (np.abs(gpu_img - ref).max() / np.abs(ref).max())),
                "numpy_oracle_s": round(np_dt, 2),
            }
        except Exception as exc:      # an auxiliary leg must never cost the headline line
            result['cpu_baseline_error'] = repr(exc)[:300]


    # ---- joint-training leg (BASELINE configs[3]): deq+lin+hal + VGG16 perceptual loss, fwd+bwd+Adam, batch 32 x
    #      256x256 per GPU, ONE RCCL all-reduce(SUM) of the flat fp32 gradient per step (weak scaling) ------------
    if args.train_steps > 0:
        if rank == 0:
            progress("joint-training leg")
        del out
        torch.cuda.empty_cache()
        tg = torch.Generator().manual_seed(4 + rank)
        b, sz = args.train_batch, args.train_size

        def q(shape):
            return torch.round(torch.rand(shape, generator=tg) * 255.0) / 255.0

        clipped = q((b, sz, sz, 3))
        sat = clipped >= 1.0
        hdr_t = torch.where(sat, clipped * (1.0 + 3.0 * torch.rand((b, sz, sz, 3), generator=tg)), clipped)
        inv = torch.cumsum(torch.rand((b, 1024), generator=tg), dim=1)
        inv = (inv - inv[:, :1]) / (inv[:, -1:] - inv[:, :1])
        ds = tuple(t.cuda() for t in (q((b, sz, sz, 3)), q((b, sz, sz, 3)), clipped, hdr_t, torch.ones(b, 1, 1, 1)))
        inv = inv.cuda()
        vg = torch.Generator().manual_seed(99)
        dd = {}
        for name, cin, cout in (("conv1_1", 3, 64), ("conv1_2", 64, 64), ("conv2_1", 64, 128), ("conv2_2", 128, 128),
                                ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256)):
            lim = (6.0 / (9 * cin + 9 * cout)) ** 0.5
            dd[name] = [((torch.rand((3, 3, cin, cout), generator=vg) * 2 - 1) * lim).numpy(), torch.zeros(cout).numpy()]
        vgg = pkg.vgg16.Vgg16(data_dict=dd)
        step = pkg.pipeline.JointTrainStep(deq, lin, hal, vgg, process_group=(dist.group.WORLD if dist is not None else None),
                                           world_size=world, multi_stream=not args.single_stream)
        step(ds, inv)                      # warm-up (allocator, kernel attributes, RCCL rings)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.train_steps):
            tout = step(ds, inv)
        barrier()
        tdt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([tdt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tdt = float(t.item())
        loss = float(tout["total"].detach().sum())
        assert loss == loss, "joint step produced NaN"
        # algorithmic FLOPs per image: fwd 102.1 + 2 x VGG 24.39, bwd 2 x 102.1 + 24.39 GF at 256^2
        gflop_img = 379.5 * (sz / 256.0) ** 2
        # the batch builder of that step (joint_training.py:26-69): exposure + noise + CRF + JPEG round trip + loss mask on the
        # device, with the libjpeg-on-host round trip the reference pays (Pillow links the same libjpeg-turbo) beside it
        cam = pkg.camera.CameraPipeline(seed=1 + rank)
        crf = torch.cumsum(torch.rand((b, 1024), generator=tg), dim=1)
        crf = ((crf - crf[:, :1]) / (crf[:, -1:] - crf[:, :1])).cuda()
        expo = (0.5 + 2.0 * torch.rand((b,), generator=tg)).cuda()
        cam(hdr_t.cuda(), crf, expo)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(5):
            cam_out = cam(hdr_t.cuda(), crf, expo)
        torch.cuda.synchronize()
        cam_ms = (time.perf_counter() - c0) / 5 * 1e3
        host_jpeg_ms = None
        if rank == 0:
            try:
                import io
                from PIL import Image
                u8 = torch.round(cam_out[0] * 255.0).to(torch.uint8).cpu().numpy()
                c0 = time.perf_counter()
                for i, qq in enumerate(pkg.camera.jpeg_qualities(b)):
                    buf = io.BytesIO()
                    Image.fromarray(u8[i]).save(buf, format="JPEG", quality=qq, subsampling=2)
                    np.array(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
                host_jpeg_ms = (time.perf_counter() - c0) * 1e3
            except ImportError:
                pass
        result["joint_train"] = {
            "workload": "BASELINE configs[3]: joint_training.py step, batch=%d x %dx%d per GPU, fp32, "
                        "1 all-reduce(SUM) of %d fp32 gradients" % (b, sz, sz, step.params.num_params),
            "ms_per_step": round(tdt / args.train_steps * 1e3, 2), "steps": args.train_steps,
            "images_per_s": round(b * world * args.train_steps / tdt, 2), "n_gpus": world, "scaling": "weak",
            "tflops_algorithmic_per_gpu": round(gflop_img * b * args.train_steps / tdt / 1e3, 2),
            "loss": round(loss, 5),
            "camera_pipeline_ms_per_batch": round(cam_ms, 3),
            "host_libjpeg_round_trip_ms_per_batch": None if host_jpeg_ms is None else round(host_jpeg_ms, 2),
        }
        if rank == 0 and world == 1 and not args.no_roofline:          # (the instrumented step all-reduces: N = 1 only)
            try:
                result["joint_train"]["roofline"] = train_roofline(K, lambda: step(ds, inv, apply=False), args.layers)
            except Exception as exc:
                result["joint_train"]["roofline_error"] = repr(exc)[:300]

    # ---- fine-tuning leg (BASELINE configs[4]): the chained deq->lin->hal->ref step of finetune_real_dataset.py on
    #      1024x1024 tiles, fp32 operands vs the fp16-MFMA-operand conv path (fp32 master weights / accumulation) ----
    if args.finetune_steps > 0:
        tout = step = None
        torch.cuda.empty_cache()
        fg = torch.Generator().manual_seed(5 + rank)
        b, sz = args.finetune_batch, args.finetune_size
        f_ldr = (torch.round(torch.rand((b, sz, sz, 3), generator=fg) * 255.0) / 255.0).cuda()
        f_hdr = torch.rand((b, sz, sz, 3), generator=fg) * 1.5
        f_hdr = (f_hdr / (1e-6 + f_hdr.mean(dim=(1, 2, 3), keepdim=True)) * 0.5).cuda()
        gflop_img = 5360.0 * (sz / 1024.0) ** 2         # fwd 1785 GF/img, fwd+bwd ~3x
        leg = {"workload": "BASELINE configs[4]: finetune_real_dataset.py step (deq+lin+hal+ref, fwd+bwd+Adam), batch=%d x "
                           "%dx%d tiles per GPU; fp16 = NATIVE fp16 conv path: fp16 feature maps in HBM and LDS, "
                           "v_mfma_f32_16x16x32_f16 in every conv fwd/dgrad/wgrad, fp16 BatchNorm / pooling / resize passes, "
                           "fp32 master weights, parameter gradients and accumulation" % (b, sz, sz),
               "n_gpus": world, "scaling": "weak", "steps": args.finetune_steps}
        for prec in [p for p in args.finetune_prec.split(",") if p]:
            if rank == 0:
                progress("fine-tuning leg, %s" % prec)
            torch.manual_seed(777)
            nets4 = [pkg.dequantization_net.model(), pkg.linearization_net.model(), pkg.hallucination_net.model(),
                     pkg.refinement_net.model()]
            fstep = pkg.pipeline.FinetuneStep(*nets4, precision=prec, loss_scale=1.0 if prec == "fp32" else 0.25,
                                              process_group=(dist.group.WORLD if dist is not None else None), world_size=world)
            fstep(f_ldr, f_hdr)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.finetune_steps):
                fout = fstep(f_ldr, f_hdr)
            barrier()
            fdt = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([fdt], device="cuda", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                fdt = float(t.item())
            floss = float(fout["loss_sum"].detach().sum())
            assert floss == floss, "fine-tuning step produced NaN (%s)" % prec
            leg[prec] = {"ms_per_step": round(fdt / args.finetune_steps * 1e3, 2),
                         "images_per_s": round(b * world * args.finetune_steps / fdt, 3),
                         "tflops_algorithmic_per_gpu": round(gflop_img * b * args.finetune_steps / fdt / 1e3, 2),
                         "loss_sum": round(floss, 3), "skipped_steps": fstep.skipped_steps}
            if prec == "fp32" and rank == 0 and world == 1 and not args.no_roofline:
                try:
                    leg[prec]["roofline"] = train_roofline(K, lambda: fstep(f_ldr, f_hdr, apply=False), args.layers)
                except Exception as exc:
                    leg[prec]["roofline_error"] = repr(exc)[:300]
            if prec == "fp16" and rank == 0 and world == 1 and not args.no_roofline:      # the instrumented step all-reduces: N = 1 only
                try:
                    leg[prec]["roofline"] = fp16_roofline(K, fstep, f_ldr, f_hdr)
                except Exception as exc:
                    leg[prec]["roofline_error"] = repr(exc)[:300]
            del fstep, nets4, fout
            torch.cuda.empty_cache()
        result["finetune"] = leg

    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

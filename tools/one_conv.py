#!/usr/bin/env python3
"""Run ONE conv layer of the hot path repeatedly (for rocprofv3 --pmc / kernel-trace).

    python3 tools/one_conv.py [--n 16] [--hw 128] [--cin 512] [--cout 256] [--k 3] [--stride 1] [--c2 0] [--reps 20]
Prints the HIP-event average time and TFLOP/s.
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--hw", type=int, default=128)
    ap.add_argument("--cin", type=int, default=512)
    ap.add_argument("--c2", type=int, default=0)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--res", action="store_true", help="fused residual add + BN scale/shift in the epilogue")
    ap.add_argument("--algo", type=int, default=0, help="SHDR_ALGO_* (4: fp16 MFMA operands, 5: bf16)")
    a = ap.parse_args()
    K = importlib.import_module("singlehdr-tf2_amd")._ops
    torch.manual_seed(0)
    x = torch.randn(a.n, a.hw, a.hw, a.cin, device="cuda")
    x2 = torch.randn(a.n, a.hw, a.hw, a.c2, device="cuda") if a.c2 else None
    w = torch.randn(a.k, a.k, a.cin + a.c2, a.cout, device="cuda") * 0.02
    b = torch.randn(a.cout, device="cuda")
    ho0 = -(-a.hw // a.stride)
    extra = dict(residual=torch.randn(a.n, ho0, ho0, a.cout, device="cuda"), scale=torch.rand(a.cout, device="cuda"), shift=torch.randn(a.cout, device="cuda"), act2=K.ACT_RELU) if a.res else {}
    for _ in range(3):
        y = K.conv2d(x, w, b, stride=a.stride, x2=x2, act1=K.ACT_NONE if a.res else K.ACT_RELU, algo=a.algo, **extra)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.reps):
        y = K.conv2d(x, w, b, stride=a.stride, x2=x2, act1=K.ACT_NONE if a.res else K.ACT_RELU, algo=a.algo, **extra)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    ho = -(-a.hw // a.stride)
    fl = 2.0 * a.n * ho * ho * (a.cin + a.c2) * a.cout * a.k * a.k
    if a.algo in (4, 5):
        K.WINOGRAD = False
        ref = K.conv2d(x, w, b, stride=a.stride, x2=x2, act1=K.ACT_RELU)
        print("  rel err vs fp32 path: %.3g" % float((y - ref).abs().max() / ref.abs().max()))
    print("conv %dx%dx%d %d+%d->%d k%d s%d: %.4f ms  %.2f TFLOP/s  (y mean %.4g)"
          % (a.n, a.hw, a.hw, a.cin, a.c2, a.cout, a.k, a.stride, ms, fl / ms / 1e9, float(y.mean())))


if __name__ == "__main__":
    main()

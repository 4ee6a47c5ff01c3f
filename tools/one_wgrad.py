#!/usr/bin/env python3
"""Time the weight-gradient kernel on ONE layer shape:  python3 tools/one_wgrad.py --n 32 --hw 64 --cin 256 --cout 256 --k 3"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--hw", type=int, default=64)
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--prec", default="fp32")
    a = ap.parse_args()
    K = importlib.import_module("singlehdr-tf2_amd")._ops
    torch.manual_seed(0)
    x = torch.randn(a.n, a.hw, a.hw, a.cin, device="cuda")
    ho = -(-a.hw // a.stride)
    dz = torch.randn(a.n, ho, ho, a.cout, device="cuda")
    shape = (a.k, a.k, a.cin, a.cout)
    with K.precision(a.prec):
        for _ in range(2):
            dw = K.conv2d_wgrad(x, None, dz, shape, a.stride)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps):
            dw = K.conv2d_wgrad(x, None, dz, shape, a.stride)
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    fl = 2.0 * a.n * ho * ho * a.cin * a.cout * a.k * a.k
    print("wgrad %dx%dx%d %d->%d k%d %s: %.4f ms  %.2f TFLOP/s  (|dw| %.4g)"
          % (a.n, a.hw, a.hw, a.cin, a.cout, a.k, a.prec, ms, fl / ms / 1e9, float(dw.abs().mean())))


if __name__ == "__main__":
    main()

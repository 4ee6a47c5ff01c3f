#!/usr/bin/env python3
"""L2-relative error of the input gradient of one wide 3x3 layer vs float64, split-operand kernel vs the exact-fp32 kernel, for
output gradients with a heavy-tailed magnitude distribution (as in the training steps)"""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
K = importlib.import_module("singlehdr-tf2_amd")._ops
os.environ["SHDR_X3_MIN_BLOCKS"] = "1"
torch.manual_seed(0)
n, h, w, cin, cout = 2, 64, 64, 128, 128
wt = (torch.randn(3, 3, cin, cout, device="cuda") / (9 * cin) ** 0.5)
wf = wt.double().flip(0, 1).permute(2, 3, 0, 1)          # conv_transpose as conv: [cin, cout, 3, 3]
for name, mk in (("N(0,1)", lambda z: z), ("N(0,1) * 1e-6", lambda z: z * 1e-6), ("log-normal tails, 50% zeros, * 1e-5",
                 lambda z: z * torch.exp(3 * torch.randn_like(z)) * (torch.rand_like(z) > 0.5) * 1e-5)):
    dz = mk(torch.randn(n, h, w, cout, device="cuda")).contiguous()
    ref = torch.nn.functional.conv2d(dz.double().permute(0, 3, 1, 2), wf, padding=1).permute(0, 2, 3, 1)
    out = {}
    for exact in (False, True):
        K.EXACT_FP32 = exact
        dx = K.conv2d_dgrad(dz, wt, (n, h, w, cin), cin, 0, 0)
        out[exact] = (float((dx.double() - ref).norm() / ref.norm()), float((dx.double() - ref).abs().max() / ref.abs().max()))
    K.EXACT_FP32 = False
    print("%-40s x3: L2 %.2e max %.2e    exact fp32: L2 %.2e max %.2e" % (name, out[False][0], out[False][1], out[True][0], out[True][1]))

#!/usr/bin/env python3
"""HBM rate of the elementwise / reduction kernels of the training tape at the joint-training shapes (batch 32 x 256^2)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("singlehdr-tf2_amd")._ops


def t(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


HALF = "--half" in sys.argv          # the fp16 twins at the fine-tuning shapes (batch 4 x 1024^2)
SHAPES = [(4, 1024, 1024, 64), (4, 512, 512, 128), (4, 512, 512, 64), (4, 256, 256, 256), (4, 128, 128, 512), (4, 64, 64, 512),
          (4, 1024, 1024, 16)] if HALF else [(32, 128, 128, 64), (32, 64, 64, 256), (32, 32, 32, 512), (32, 256, 256, 64), (32, 16, 16, 1024)]
for shape in SHAPES:
    x = torch.randn(*shape, device="cuda"); dy = torch.randn_like(x); c = shape[-1]
    if HALF:
        x, dy = x.half(), dy.half()
    g, b = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda")
    mb = x.numel() * x.element_size() / 1e6
    ms = t(lambda: K.bn_stats(x)); print("%-22s bn_stats        %.3f ms  %.2f TB/s" % (shape, ms, mb / ms / 1e3))
    mean, var = K.bn_stats(x)
    ms = t(lambda: K.bn_train_apply(x, mean, var, g, b, 1e-3, True)); print("%-22s bn_train_apply  %.3f ms  %.2f TB/s" % (shape, ms, 2 * mb / ms / 1e3))
    y = K.bn_train_apply(x, mean, var, g, b, 1e-3, True)
    ms = t(lambda: K.bn_bwd(dy, x, y, mean, var, g, 1e-3)); print("%-22s bn_bwd          %.3f ms  %.2f TB/s (5 passes + 1 write)" % (shape, ms, 6 * mb / ms / 1e3))
    ms = t(lambda: K.act_bwd_bias(dy, y, K.ACT_RELU)); print("%-22s act_bwd_bias    %.3f ms  %.2f TB/s" % (shape, ms, 3 * mb / ms / 1e3))
    ms = t(lambda: K.act_bwd(dy, y, K.ACT_RELU)); print("%-22s act_bwd         %.3f ms  %.2f TB/s" % (shape, ms, 3 * mb / ms / 1e3))
    ms = t(lambda: K.add(dy, y)); print("%-22s add             %.3f ms  %.2f TB/s" % (shape, ms, 3 * mb / ms / 1e3))
    ms = t(lambda: torch.add(dy, y)); print("%-22s torch.add       %.3f ms  %.2f TB/s" % (shape, ms, 3 * mb / ms / 1e3))

#!/usr/bin/env python3
"""Gradient of the joint training step with the split-operand kernel in forward + dgrad vs the exact-fp32 kernels (same weights, same
batch): how far apart are they, per network, next to the run-to-run spread of the exact path (atomics)?  Also the magnitudes of dz."""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
shdr = importlib.import_module("singlehdr-tf2_amd")
K = shdr._ops
torch.manual_seed(0)
os.environ["SHDR_X3_MIN_BLOCKS"] = "1"
B, S = 8, 256
deq, lin, hal = shdr.dequantization_net.model(), shdr.linearization_net.model(), shdr.hallucination_net.model()
vg = torch.Generator().manual_seed(99)
dd = {}
for name, cin, cout in (("conv1_1", 3, 64), ("conv1_2", 64, 64), ("conv2_1", 64, 128), ("conv2_2", 128, 128), ("conv3_1", 128, 256),
                        ("conv3_2", 256, 256), ("conv3_3", 256, 256)):
    lim = (6.0 / (9 * cin + 9 * cout)) ** 0.5
    dd[name] = [((torch.rand((3, 3, cin, cout), generator=vg) * 2 - 1) * lim).numpy(), torch.zeros(cout).numpy()]
vgg = shdr.vgg16.Vgg16(data_dict=dd)
step = shdr.pipeline.JointTrainStep(deq, lin, hal, vgg)
g = torch.Generator(device="cuda").manual_seed(1)
q = lambda: (torch.rand((B, S, S, 3), device="cuda", generator=g) * 255).round() / 255
clipped = q()
hdr = clipped * torch.where(clipped >= 0.99, 1 + 3 * torch.rand((B, S, S, 3), device="cuda", generator=g), torch.ones_like(clipped))
inv = torch.cumsum(torch.rand((B, 1024), device="cuda", generator=g), 1)
inv = (inv - inv[:, :1]) / (inv[:, -1:] - inv[:, :1])
mask = torch.ones((B, 1, 1, 1), device="cuda")
batch = (q(), q(), clipped, hdr, mask)
dzmax = []
orig = K.conv2d_dgrad
def spy(dz, *a, **k):
    dzmax.append(float(dz.abs().max()))
    return orig(dz, *a, **k)
K.conv2d_dgrad = spy
def grads(exact, exact_bwd=None):
    K.EXACT_FP32 = exact
    if exact_bwd is None:
        step(batch, inv, apply=False)
    else:                                  # forward with one kernel set, backward with the other: isolates mask flips from dgrad rounding
        step.params.zero_grad()
        out = step.losses(batch, inv)
        K.EXACT_FP32 = exact_bwd
        out["objective"].backward()
    return step.params.grad.clone()
gx = grads(False); n_dz = len(dzmax)
g_fe_bx = grads(True, False)               # exact forward, x3 dgrad
g_fx_be = grads(False, True)               # x3 forward, exact dgrad
K.EXACT_FP32 = True
step.params.zero_grad(); out = step.losses(batch, inv)
os.environ["SHDR_NO_WINOGRAD"] = "1"       # exact forward (Winograd), backward on the direct fp32-MFMA kernels: two EXACT-fp32 dgrad kernels compared
out["objective"].backward(); del os.environ["SHDR_NO_WINOGRAD"]
g_fe_bd = step.params.grad.clone()
ge1 = grads(True); ge2 = grads(True)
K.conv2d_dgrad = orig
rel = lambda a, b: float((a - b).norm() / b.norm())
print("exact forward + DIRECT fp32 dgrad vs exact (Winograd fp32 dgrad): %.2e" % float((g_fe_bd - ge1).norm() / ge1.norm()))
print("exact forward + x3 dgrad vs exact: %.2e      x3 forward + exact dgrad vs exact: %.2e" % (float((g_fe_bx - ge1).norm() / ge1.norm()), float((g_fx_be - ge1).norm() / ge1.norm())))
print("flat gradient: |x3 - exact| / |exact| = %.2e   exact run-to-run = %.2e   (norm %.3e)" % (rel(gx, ge1), rel(ge2, ge1), float(ge1.norm())))
off = 0
for name, m in (("deq", deq), ("lin", lin), ("hal", hal)):
    n = sum(v.numel() for v in m.trainable_variables)
    print("  %s: x3 vs exact %.2e   exact run-to-run %.2e" % (name, rel(gx[off:off + n], ge1[off:off + n]), rel(ge2[off:off + n], ge1[off:off + n])))
    off += n
d = np.array(dzmax[:n_dz])
print("max|dz| over the %d dgrad calls of one step: min %.2e  median %.2e  max %.2e" % (len(d), d.min(), np.median(d), d.max()))

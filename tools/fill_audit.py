#!/usr/bin/env python3
"""Which host-side allocations launch a fill kernel in one joint training step (torch.zeros / zeros_like / full / Tensor.zero_):
python tools/fill_audit.py   -> (caller, shape) counts of one warm step at the bench size (batch 32 x 256^2)"""
import collections, importlib, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("singlehdr-tf2_amd")
b, sz = 32, 256
deq, lin, hal = pkg.dequantization_net.model(), pkg.linearization_net.model(), pkg.hallucination_net.model()
tg = torch.Generator().manual_seed(4)


def q(shape):
    return torch.round(torch.rand(shape, generator=tg) * 255.0) / 255.0


clipped = q((b, sz, sz, 3))
hdr_t = torch.where(clipped >= 1.0, clipped * (1.0 + 3.0 * torch.rand((b, sz, sz, 3), generator=tg)), clipped)
inv = torch.cumsum(torch.rand((b, 1024), generator=tg), dim=1)
inv = ((inv - inv[:, :1]) / (inv[:, -1:] - inv[:, :1])).cuda()
ds = tuple(t.cuda() for t in (q((b, sz, sz, 3)), q((b, sz, sz, 3)), clipped, hdr_t, torch.ones(b, 1, 1, 1)))
vg = torch.Generator().manual_seed(99)
dd = {}
for name, cin, cout in (("conv1_1", 3, 64), ("conv1_2", 64, 64), ("conv2_1", 64, 128), ("conv2_2", 128, 128),
                        ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256)):
    lim = (6.0 / (9 * cin + 9 * cout)) ** 0.5
    dd[name] = [((torch.rand((3, 3, cin, cout), generator=vg) * 2 - 1) * lim).numpy(), torch.zeros(cout).numpy()]
step = pkg.pipeline.JointTrainStep(deq, lin, hal, pkg.vgg16.Vgg16(data_dict=dd))
for _ in range(2):
    step(ds, inv)
torch.cuda.synchronize()
seen = collections.Counter()


def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "singlehdr-tf2_amd" in fr.filename:
            return "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
    return "?"


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(*a, **kw):
        r = orig(*a, **kw)
        if isinstance(r, torch.Tensor) and r.is_cuda:
            seen[(name, where(), tuple(r.shape))] += 1
        return r
    setattr(mod, name, f)


for n in ("zeros", "zeros_like", "full", "ones"):
    wrap(torch, n)
zero_ = torch.Tensor.zero_


def zero_logged(self):
    seen[("zero_", where(), tuple(self.shape))] += 1
    return zero_(self)


torch.Tensor.zero_ = zero_logged
pkg._ops.RANGE_MISSES.clear()
step(ds, inv)
torch.cuda.synchronize()
for k, v in sorted(pkg._ops.RANGE_MISSES.items(), key=lambda kv: -kv[1]):
    print("range measured %d x: %s" % (v, k))
tot = 0
for (k, v) in sorted(seen.items(), key=lambda kv: -kv[1]):
    tot += v
    print("%4d x %-10s %-48s %s" % (v, k[0], k[1], k[2]))
print("total", tot)

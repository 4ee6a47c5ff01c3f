import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
shdr = importlib.import_module("singlehdr-tf2_amd"); K = shdr._ops
torch.manual_seed(0)
deq = shdr.dequantization_net.model()
orig = K.conv2d
def spy(x, w, bias=None, **kw):
    y = orig(x, w, bias, **kw)
    c2 = 0 if kw.get("x2") is None else kw["x2"].shape[3]
    plan = K.conv2d_plan(tuple(x.shape), tuple(w.shape), c2=c2, has_residual=kw.get("residual") is not None, cout_valid=kw.get("cout_valid"))
    print("%-5s x %s w %s c2 %d -> finite %s  max %.3e" % (plan, tuple(x.shape), tuple(w.shape), c2, bool(torch.isfinite(y).all()), float(y.abs().max())), flush=True)
    return y
K.conv2d = spy
for n, s in ((1, 512), (2, 512)):
    x = (torch.rand(n, s, s, 3, device="cuda") * 255).round() / 255
    with torch.no_grad():
        y = deq(x, training=False)
    print("N", n, s, "finite", bool(torch.isfinite(y).all()))

import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
shdr = importlib.import_module("singlehdr-tf2_amd")
K = shdr._ops
torch.manual_seed(0)
for (n, h, w, cx, cz, cout_valid, k) in [(2, 12, 20, 16, 8, 3, 3), (2, 12, 20, 16, 16, 16, 3), (1, 8, 8, 16, 8, 8, 1), (1, 8, 32, 16, 8, 3, 1)]:
    x = torch.randn(n, h, w, cx).half().cuda()
    dz = torch.zeros(n, h, w, cz).half()
    dz[..., :cout_valid] = torch.randn(n, h, w, cout_valid).half()
    dz = dz.cuda()
    coutw = 16
    dw = K.conv2d_wgrad_h(x, None, dz, (k, k, cx, coutw), 1, 1.0, cout_valid=cout_valid)
    # reference: dW[kh,kw,ci,co] = sum x[p+tap, ci] dz[p, co]
    xp = torch.nn.functional.pad(x.float().cpu(), (0, 0, k // 2, k // 2, k // 2, k // 2))
    ref = torch.zeros(k, k, cx, coutw)
    for a in range(k):
        for b in range(k):
            ref[a, b, :, :cz] = torch.einsum("nhwc,nhwd->cd", xp[:, a:a + h, b:b + w, :], dz.float().cpu())
    ref[..., cout_valid:] = 0
    err = (dw.cpu() - ref).abs()
    print((n, h, w, cx, cz, cout_valid, k), "max err", float(err.max()), "ref max", float(ref.abs().max()))
    if err.max() > 0.05:
        bad = (err > 0.05).nonzero()
        print("bad count", len(bad), "of", int((ref != 0).sum()), "first", bad[:10].tolist())
        print("dw[0,0,:4,:4]", dw.cpu()[0, 0, :4, :4], "\nref", ref[0, 0, :4, :4])

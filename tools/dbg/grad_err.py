"""prints the actual whole-net gradient errors (relative L2 per variable, worst) of lin / hal vs the float64 reference"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch_ref as R
from oracle import nets
shdr = importlib.import_module("singlehdr-tf2_amd")
K = shdr._ops
table = np.load(os.path.join(ROOT, "singlehdr-tf2_amd", "data", "invemor_g0_hinv11.npy"))
def dev(x, g=False): return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda().requires_grad_(g)
def host(t): return t.detach().cpu().numpy()
def q(rng, shape): return np.round(rng.random(shape) * 255.0) / 255.0
def report(tag, m, tp):
    named = [(n, t) for n, t, tr in m.named_weights() if tr]
    tot_d = tot_r = 0.0
    worst = ("", 0.0)
    for n, t in named:
        ref = tp[n].grad.numpy(); d = host(t.grad).astype(np.float64) - ref
        tot_d += (d ** 2).sum(); tot_r += (ref ** 2).sum()
        l2 = np.linalg.norm(d) / max(np.linalg.norm(ref), 1e-30)
        if l2 > worst[1] and np.linalg.norm(ref) > 1e-3 * np.sqrt(tot_r + 1e-30): worst = (n, l2)
    print("%-28s whole-net rel L2 %.3e   worst variable %s %.3e" % (tag, np.sqrt(tot_d / tot_r), worst[0], worst[1]), flush=True)
for seed in (62, 162):
    p = nets.init_params(nets.lin_spec(), seed); m = shdr.linearization_net.model().load_numpy(p)
    rng = np.random.default_rng(seed); x = q(rng, (2, 64, 64, 3)); inv = np.cumsum(rng.random((2, 1024)), axis=1); inv /= inv[:, -1:]
    tp = R.params_to_torch(p); tinv = R.lin_forward(tp, R.T(x), table, True)
    ((tinv - R.T(inv)) ** 2).mean(dim=1).sum().backward()
    pred = m(dev(x), training=True); K.diff_loss(pred, dev(inv), 0).sum().backward()
    report("lin seed %d (through _increase)" % seed, m, tp)
    # the same net, loss on the feature vector BEFORE the CRF head's min(): isolates the trunk
for seed in (63, 163):
    p = nets.init_params(nets.hal_spec(), seed); m = shdr.hallucination_net.model().load_numpy(p)
    rng = np.random.default_rng(seed); x, tgt = q(rng, (2, 64, 64, 3)), rng.random((2, 64, 64, 3))
    tp = R.params_to_torch(p); ty = R.hal_forward(tp, R.T(x), True)
    (ty - R.T(tgt)).abs().mean(dim=(1, 2, 3)).sum().backward()
    y = m(dev(x), training=True); K.diff_loss(y, dev(tgt), 1).sum().backward()
    report("hal seed %d (L1 loss)" % seed, m, tp)
    for _, t, _ in m.named_weights(): t.grad = None
    tp = R.params_to_torch(p); ty = R.hal_forward(tp, R.T(x), True)
    ((ty - R.T(tgt)) ** 2).mean(dim=(1, 2, 3)).sum().backward()
    y = m(dev(x), training=True); K.diff_loss(y, dev(tgt), 0).sum().backward()
    report("hal seed %d (L2 loss)" % seed, m, tp)

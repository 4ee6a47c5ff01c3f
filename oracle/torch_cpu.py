"""Torch-CPU fp32 channels-last restatement of the inference path -- TEST INFRASTRUCTURE and the `cpu_baseline` of bench.py.

This is the "CPU restatement (proxy for TF2-CPU)" of BASELINE.md section 4: TensorFlow is not installable here, so the reported
CPU baseline is this module -- the same deq -> clip -> lin -> apply_rf -> alpha -> hal -> blend path as
test_real_refinement.py:86-105 of the reference, on torch's CPU kernels (oneDNN convolutions on channels-last tensors, all
host cores), fp32, TF-style explicit SAME padding, identical inputs and weights.  The product never imports it.
It is pinned to the NumPy oracle (oracle/nets.py) by tests/test_oracle.py::test_torch_cpu_baseline_equals_numpy_oracle.

Reference lines restated: dequantization_net.py:4-65, linearization_net.py:6-118,173-196,310-350,368-392,
hallucination_net.py:43-190, tf_utils.py:54-105, test_real_refinement.py:86-105.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops as nops


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _cl(x):
    return x.contiguous(memory_format=torch.channels_last)


class Net:
    """parameters {name: ndarray} (oracle naming, HWIO kernels) -> OIHW channels-last torch tensors, converted once"""

    def __init__(self, params):
        self.p = {}
        for k, v in params.items():
            t = _t(v)
            if k.endswith(".kernel") and t.dim() == 4:
                t = _cl(t.permute(3, 2, 0, 1))
            self.p[k] = t

    def conv(self, name, x, stride=1):
        """Conv2D SAME (TF: the extra padding cell goes to the bottom / right)"""
        w = self.p[name + ".kernel"]
        kh, kw = w.shape[2], w.shape[3]
        _, pt, pb = nops.same_pad(x.shape[2], kh, stride)
        _, pl, pr = nops.same_pad(x.shape[3], kw, stride)
        if pt != pb or pl != pr:
            x = F.pad(x, (pl, pr, pt, pb))
            pad = 0
        else:
            pad = (pt, pl)
        return F.conv2d(x, w, self.p.get(name + ".bias"), stride=stride, padding=pad)

    def bn(self, name, x, eps=1e-3):
        scale = self.p[name + ".gamma"] / torch.sqrt(self.p[name + ".moving_variance"] + eps)
        shift = self.p[name + ".beta"] - self.p[name + ".moving_mean"] * scale
        return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


def lrelu(x):
    return F.leaky_relu(x, 0.1)


def resize2x(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


def deq_forward(net, x):
    def down(n, t):
        t = F.avg_pool2d(t, 2)
        return lrelu(net.conv(n + ".conv2", lrelu(net.conv(n + ".conv1", t))))

    def up(n, t, skip):
        t = lrelu(net.conv(n + ".conv1", resize2x(t)))
        return lrelu(net.conv(n + ".conv2", torch.cat([t, skip], 1)))

    t = lrelu(net.conv("conv1", x))
    s1 = lrelu(net.conv("conv2", t))
    s2 = down("d2", s1)
    s3 = down("d3", s2)
    s4 = down("d4", s3)
    t = down("enc", s4)
    t = up("u4", t, s4)
    t = up("u3", t, s3)
    t = up("u2", t, s2)
    t = up("u1", t, s1)
    return x + torch.tanh(net.conv("out", t))


def lin_frontend(img):
    """[n,3,h,w] -> [n,93,h,w]: image, sobel (REFLECT, channel = c*2 + {dy,dx}), soft histograms B = 4, 8, 16"""
    n, c, h, w = img.shape
    xp = F.pad(img, (1, 1, 1, 1), mode="reflect")
    ky = torch.tensor([[-1., -2., -1.], [0., 0., 0.], [1., 2., 1.]])
    kern = torch.stack([ky, ky.T])[:, None]
    edges = F.conv2d(xp.reshape(n * c, 1, h + 2, w + 2), kern).reshape(n, 2 * c, h, w)
    feats = [img, edges]
    for B in (4, 8, 16):
        for i in range(1, B + 1):
            d = (img - np.float32(2 * i - 1) / np.float32(2 * B)).abs()
            feats.append(torch.where(d < np.float32(1.0 / B), 1.0 - d * B, torch.zeros_like(d)))
    return _cl(torch.cat(feats, 1))


def lin_forward(net, img, table):
    q = "crf_feature_net."

    def res1(n, t, s):
        b1 = net.bn(n + ".norm1", net.conv(n + ".conv1", t, s))
        a = torch.relu(net.bn(n + ".norm2", net.conv(n + ".conv2", t, s)))
        a = torch.relu(net.bn(n + ".norm3", net.conv(n + ".conv3", a)))
        return torch.relu(b1 + net.bn(n + ".norm4", net.conv(n + ".conv4", a)))

    def res2(n, t):
        a = torch.relu(net.bn(n + ".norm1", net.conv(n + ".conv1", t)))
        a = torch.relu(net.bn(n + ".norm2", net.conv(n + ".conv2", a)))
        return torch.relu(t + net.bn(n + ".norm3", net.conv(n + ".conv3", a)))

    t = torch.relu(net.bn(q + "norm1", net.conv(q + "conv1", lin_frontend(img), 2)))
    _, pt, pb = nops.same_pad(t.shape[2], 3, 2)
    _, pl, pr = nops.same_pad(t.shape[3], 3, 2)
    t = F.max_pool2d(F.pad(t, (pl, pr, pt, pb), value=float("-inf")), 3, 2)
    t = res1(q + "res1", t, 1)
    t = res2(q + "res2", t)
    t = res2(q + "res3", t)
    t = res1(q + "res4", t, 2)
    t = res2(q + "res5", t)
    feat = t.mean(dim=(2, 3))
    wts = feat @ net.p["ae_invcrf_decode_net.fc.kernel"] + net.p["ae_invcrf_decode_net.fc.bias"]
    tab = _t(table)
    rf = tab[:, 0][None] + wts @ tab[:, 1:].T
    g = rf[:, 1:] - rf[:, :-1]
    ng = g + torch.relu(-g.min(dim=-1, keepdim=True).values)
    ng = ng / ng.sum(dim=-1, keepdim=True)
    return F.pad(torch.cumsum(ng, dim=-1), (1, 0))


def apply_rf(x, rf):
    b, k = rf.shape
    y = (k - 1) * x.reshape(b, -1)
    y0 = torch.floor(y)
    i0 = y0.long().clamp(0, k - 1)
    i1 = (y0 + 1).long().clamp(0, k - 1)
    return ((y0 + 1 - y) * torch.gather(rf, 1, i0) + (y - y0) * torch.gather(rf, 1, i1)).reshape(x.shape)


def hal_forward(net, x):
    mean = torch.tensor([103.939, 116.779, 123.68]).view(1, 3, 1, 1)
    bgr = _cl(x.flip(1) * 255.0 - mean)

    def down(n, t, k):
        for i in range(1, k + 1):
            t = torch.relu(net.conv("%s.conv%d" % (n, i), t))
        return F.max_pool2d(t, 2), t

    def up(n, t):
        t = torch.relu(net.conv(n + ".conv1", resize2x(t)))
        return torch.relu(net.bn(n + ".norm1", t))

    def skip(n, t, sk):
        return net.conv(n + ".conv1", torch.cat([t, sk * np.float32(1.0 / 255)], 1))

    t, d1 = down("d1", bgr, 2)
    t, d2 = down("d2", t, 2)
    t, d3 = down("d3", t, 3)
    t, d4 = down("d4", t, 3)
    t, d5 = down("d5", t, 3)
    t = torch.relu(net.bn("norm1", net.conv("conv1", t)))
    t = skip("s5", up("u5", t), d5)
    t = skip("s4", up("u4", t), d4)
    t = skip("s3", up("u3", t), d3)
    t = skip("s2", up("u2", t), d2)
    t = skip("s1", up("u1", t), d1)
    t = torch.relu(net.bn("norm2", net.conv("conv2", t)))
    return torch.relu(skip("s0", t, bgr))


def inference(nets, ldr_nhwc, table, thr=0.12):
    """test_real_refinement.py:86-105 without the Refinement-Net: NHWC ndarray in -> NHWC ndarray out.  `nets` = {"deq", "lin",
    "hal"} of `Net`."""
    with torch.no_grad():
        x = _cl(_t(ldr_nhwc).permute(0, 3, 1, 2))
        c = deq_forward(nets["deq"], x).clamp(0.0, 1.0)
        invcrf = lin_forward(nets["lin"], c, table)
        b = apply_rf(c, invcrf)
        hal = hal_forward(nets["hal"], b)
        alpha = ((b.max(dim=1, keepdim=True).values - 1.0 + thr).clamp(min=0.0) / thr).clamp(max=1.0)
        a = b + alpha * hal.flip(1)
        return a.permute(0, 2, 3, 1).contiguous().numpy()

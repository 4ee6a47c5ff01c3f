"""Torch-CPU float64 restatement of the hot path WITH autograd -- test infrastructure.

The NumPy oracle (oracle/) has no backward; gradients are checked against this
module instead.  It is pinned by tests/test_oracle_grad.py: its forward must
agree with the NumPy oracle to 1e-10 on the same parameters, so the gradients
it produces are gradients of the oracle's function.  TF semantics (SAME
padding, half-pixel bilinear, biased batch variance, ...) as in oracle/ops.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import ops as nops

DT = torch.float64


def T(x, grad=False):
    return torch.tensor(np.asarray(x), dtype=DT, requires_grad=grad)


def params_to_torch(p, grad=True):
    return {k: T(v, grad and not k.endswith(("moving_mean", "moving_variance"))) for k, v in p.items()}


# A float64 MODEL of the native-fp16 path (tests/test_gpu_fp16_oracle.py): with FP16_STORAGE on, every feature map that path keeps in
# fp16 (conv outputs with more than 3 channels, BatchNorm outputs, average pools, bilinear resizes, the front end, the vgg-style
# preprocessed input) is rounded to fp16 on its way out, straight-through for the gradient.  Arithmetic stays float64; 3-channel
# images, parameters and statistics stay unrounded -- as in the product.  With it the relu / max-pool / clip decisions of the
# reference are taken on the same rounded values the HIP forward sees, so the comparison no longer carries the mask flips of an
# unrounded reference.
FP16_STORAGE = False


class fp16_storage:
    def __enter__(self):
        global FP16_STORAGE
        self._prev, FP16_STORAGE = FP16_STORAGE, True

    def __exit__(self, *exc):
        global FP16_STORAGE
        FP16_STORAGE = self._prev
        return False


def _q(x):
    if not FP16_STORAGE or x.shape[-1] <= 3:
        return x
    return x + (x.detach().half().double() - x.detach())


def _qi(x):
    """an image entering a network as an fp16 feature map (zero-padded to 8 / 16 channels in the product)"""
    return x + (x.detach().half().double() - x.detach()) if FP16_STORAGE else x


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def conv2d(x, w, b=None, stride=1):
    kh, kw = w.shape[0], w.shape[1]
    _, pt, pb = nops.same_pad(x.shape[1], kh, stride)
    _, pl, pr = nops.same_pad(x.shape[2], kw, stride)
    y = F.conv2d(F.pad(_nchw(x), (pl, pr, pt, pb)), w.permute(3, 2, 0, 1), b, stride=stride)
    return _q(_nhwc(y))


def bn(p, name, x, training, eps=1e-3):
    g, b = p[name + ".gamma"], p[name + ".beta"]
    if training:
        mean = x.mean(dim=(0, 1, 2))
        var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
    else:
        mean, var = p[name + ".moving_mean"], p[name + ".moving_variance"]
    return _q((x - mean) / torch.sqrt(var + eps) * g + b)


def lrelu(x):
    return torch.where(x >= 0, x, 0.1 * x)


def avg_pool2(x):
    return _q(_nhwc(F.avg_pool2d(_nchw(x), 2)))


def max_pool(x, k, s):
    _, pt, pb = nops.same_pad(x.shape[1], k, s)
    _, pl, pr = nops.same_pad(x.shape[2], k, s)
    return _nhwc(F.max_pool2d(F.pad(_nchw(x), (pl, pr, pt, pb), value=float("-inf")), k, s))


def resize2x(x):
    return _q(_nhwc(F.interpolate(_nchw(x), scale_factor=2, mode="bilinear", align_corners=False)))


def _c(p, name, x, stride=1):
    return conv2d(x, p[name + ".kernel"], p.get(name + ".bias"), stride)


def deq_forward(p, x):
    def down(n, t):
        t = avg_pool2(t)
        return lrelu(_c(p, n + ".conv2", lrelu(_c(p, n + ".conv1", t))))

    def up(n, t, skip):
        t = lrelu(_c(p, n + ".conv1", resize2x(t)))
        return lrelu(_c(p, n + ".conv2", torch.cat([t, skip], -1)))

    t = lrelu(_c(p, "conv1", _qi(x)))
    s1 = lrelu(_c(p, "conv2", t))
    s2 = down("d2", s1)
    s3 = down("d3", s2)
    s4 = down("d4", s3)
    t = down("enc", s4)
    t = up("u4", t, s4)
    t = up("u3", t, s3)
    t = up("u2", t, s2)
    t = up("u1", t, s1)
    return x + torch.tanh(_c(p, "out", t))


def increase(rf):
    g = rf[:, 1:] - rf[:, :-1]
    r = torch.relu(-g.min(dim=-1, keepdim=True).values)
    ng = g + r
    ng = ng / ng.sum(dim=-1, keepdim=True)
    return F.pad(torch.cumsum(ng, dim=-1), (1, 0))


def apply_rf(x, rf):
    b, k = rf.shape
    y = (k - 1) * x.reshape(b, -1)
    y0 = torch.floor(y)
    y1 = y0 + 1
    i0 = y0.long().clamp(0, k - 1)
    i1 = y1.long().clamp(0, k - 1)
    out = (y1 - y) * torch.gather(rf, 1, i0) + (y - y0) * torch.gather(rf, 1, i1)
    return out.reshape(x.shape)


def lin_frontend(img):
    """differentiable restatement of oracle.ops.lin_frontend (sobel REFLECT + soft histograms)"""
    n, h, w, c = img.shape
    xp = F.pad(_nchw(img), (1, 1, 1, 1), mode="reflect")
    ky = torch.tensor([[-1., -2., -1.], [0., 0., 0.], [1., 2., 1.]], dtype=DT)
    kern = torch.stack([ky, ky.T])[:, None]                      # [2,1,3,3]: dy, dx
    e = F.conv2d(xp.reshape(n * c, 1, h + 2, w + 2), kern).reshape(n, c, 2, h, w)
    edges = e.permute(0, 3, 4, 1, 2).reshape(n, h, w, 2 * c)    # channel = c*2 + {dy, dx}
    feats = [img, edges]
    for B in (4, 8, 16):
        for i in range(1, B + 1):
            d = (img - (2.0 * i - 1.0) / (2.0 * B)).abs()
            feats.append(torch.where(d < 1.0 / B, 1.0 - d * B, torch.zeros_like(d)))
    return _q(torch.cat(feats, -1))


def lin_forward(p, img, table, training):
    q = "crf_feature_net."
    feat_in = lin_frontend(img) if img.requires_grad else T(nops.lin_frontend(img.detach().numpy()))

    def res1(n, t, s):
        n1 = bn(p, n + ".norm1", _c(p, n + ".conv1", t, s), training)
        a2 = torch.relu(bn(p, n + ".norm2", _c(p, n + ".conv2", t, s), training))
        a3 = torch.relu(bn(p, n + ".norm3", _c(p, n + ".conv3", a2), training))
        return torch.relu(n1 + bn(p, n + ".norm4", _c(p, n + ".conv4", a3), training))

    def res2(n, t):
        a1 = torch.relu(bn(p, n + ".norm1", _c(p, n + ".conv1", t), training))
        a2 = torch.relu(bn(p, n + ".norm2", _c(p, n + ".conv2", a1), training))
        return torch.relu(t + bn(p, n + ".norm3", _c(p, n + ".conv3", a2), training))

    t = torch.relu(bn(p, q + "norm1", _c(p, q + "conv1", feat_in, 2), training))
    t = max_pool(t, 3, 2)
    t = res1(q + "res1", t, 1)
    t = res2(q + "res2", t)
    t = res2(q + "res3", t)
    t = res1(q + "res4", t, 2)
    t = res2(q + "res5", t)
    feat = t.mean(dim=(1, 2))
    w = feat @ p["ae_invcrf_decode_net.fc.kernel"] + p["ae_invcrf_decode_net.fc.bias"]
    tab = T(table)
    return increase(tab[:, 0][None, :] + w @ tab[:, 1:12].T)


def vgg_preprocess(x):
    s = x * 255.0
    return torch.stack([s[..., 2] - nops.VGG_MEAN[0], s[..., 1] - nops.VGG_MEAN[1], s[..., 0] - nops.VGG_MEAN[2]], -1)


def hal_forward(p, x, training):
    bgr = vgg_preprocess(x)

    def down(n, t, k):
        for i in range(1, k + 1):
            t = torch.relu(_c(p, "%s.conv%d" % (n, i), t))
        return max_pool(t, 2, 2), t

    def up(n, t):
        return torch.relu(bn(p, n + ".norm1", torch.relu(_c(p, n + ".conv1", resize2x(t))), training))

    def skip(n, t, sk):
        return _c(p, n + ".conv1", torch.cat([t, sk / 255.0], -1))

    t, d1 = down("d1", _qi(bgr), 2)
    t, d2 = down("d2", t, 2)
    t, d3 = down("d3", t, 3)
    t, d4 = down("d4", t, 3)
    t, d5 = down("d5", t, 3)
    t = torch.relu(bn(p, "norm1", _c(p, "conv1", t), training))
    t = skip("s5", up("u5", t), d5)
    t = skip("s4", up("u4", t), d4)
    t = skip("s3", up("u3", t), d3)
    t = skip("s2", up("u2", t), d2)
    t = skip("s1", up("u1", t), d1)
    t = torch.relu(bn(p, "norm2", _c(p, "conv2", t), training))
    return torch.relu(skip("s0", t, bgr))


def vgg_forward(p, rgb):
    t = vgg_preprocess(rgb)
    t = torch.relu(_c(p, "conv1_2", torch.relu(_c(p, "conv1_1", t))))
    p1 = max_pool(t, 2, 2)
    t = torch.relu(_c(p, "conv2_2", torch.relu(_c(p, "conv2_1", p1))))
    p2 = max_pool(t, 2, 2)
    t = torch.relu(_c(p, "conv3_3", torch.relu(_c(p, "conv3_2", torch.relu(_c(p, "conv3_1", p2))))))
    return p1, p2, max_pool(t, 2, 2)


def logc(x):
    return torch.log(1.0 + 10.0 * x) / np.log(11.0)


def alpha_mask(x, thr=0.12):
    a = x.max(dim=3, keepdim=True).values
    return torch.clamp((a - 1.0 + thr).clamp(min=0.0) / thr, max=1.0)


def tv_loss(y):
    return (y[:, 1:] - y[:, :-1]).abs().sum() / y.numel() + (y[:, :, 1:] - y[:, :, :-1]).abs().sum() / y.numel()


def joint_losses(params, vgg_params, batch, invcrf_gt, table, thr=0.12):
    """joint_training.py:137-183 with training-mode BN; every tensor float64, shapes as in the reference: the per-sample terms
    are [b,1,1,1], crf_loss is [b,1], so loss_lin and total broadcast to [b,1,b,1] (see oracle.nets.joint_losses)."""
    ldr, jpeg, clipped, hdr_t, mask = batch
    m = mask.reshape(-1, 1, 1, 1)

    def per(t):
        return t.mean(dim=(1, 2, 3), keepdim=True)
    alpha = alpha_mask(clipped, thr)
    c_pred = torch.clamp(deq_forward(params["deq"], jpeg), 0, 1)
    loss_deq = per((c_pred - ldr) ** 2) * m
    pred_invcrf = lin_forward(params["lin"], ldr, table, True)
    b_pred = apply_rf(ldr, pred_invcrf)
    crf_loss = ((pred_invcrf - invcrf_gt) ** 2).mean(dim=1, keepdim=True)
    loss_lin = (10.0 * per((b_pred - clipped) ** 2) + crf_loss) * m
    hal = hal_forward(params["hal"], clipped, True)
    a_pred = clipped + alpha * hal.flip(-1)
    ya, yh = logc(a_pred), logc(hdr_t)
    perc = 0
    for fa, fb in zip(vgg_forward(vgg_params, ya), vgg_forward(vgg_params, yh)):
        perc = perc + per((fa - fb).abs())
    l1 = per((ya - yh).abs())
    loss_hal = (l1 + 0.001 * perc + 0.1 * tv_loss(ya)) * m
    return dict(total=loss_deq + loss_lin + loss_hal, loss_deq=loss_deq, loss_lin=loss_lin, loss_hal=loss_hal,
                crf_loss=crf_loss, C_pred=c_pred, B_pred=b_pred, A_pred=a_pred)


def lin_train_loss(params, ldr, clipped, mask, invcrf_gt, table):
    """train.py:183-191: (l2 [b,1,1,1] + 0.1 * crf_loss [b,1]) * loss_mask -> [b,1,b,1], the same broadcast"""
    pred_invcrf = lin_forward(params, ldr, table, True)
    b_pred = apply_rf(ldr, pred_invcrf)
    crf_loss = ((pred_invcrf - invcrf_gt) ** 2).mean(dim=1, keepdim=True)
    l2 = ((b_pred - clipped) ** 2).mean(dim=(1, 2, 3), keepdim=True)
    return (l2 + 0.1 * crf_loss) * mask.reshape(-1, 1, 1, 1), crf_loss, b_pred


def ref_forward(p, x):
    def down(n, t):
        t = avg_pool2(t)
        return lrelu(_c(p, n + ".conv2", lrelu(_c(p, n + ".conv1", t))))

    def up(n, t, skip):
        t = lrelu(_c(p, n + ".conv1", resize2x(t)))
        return lrelu(_c(p, n + ".conv2", torch.cat([t, skip], -1)))

    t = lrelu(_c(p, "conv1", _qi(x)))
    s1 = lrelu(_c(p, "conv2", t))
    s2 = down("d2", s1)
    s3 = down("d3", s2)
    s4 = down("d4", s3)
    t = down("enc", s4)
    t = up("u4", t, s4)
    t = up("u3", t, s3)
    t = up("u2", t, s2)
    t = up("u1", t, s1)
    return torch.relu(x[..., 0:3] + _c(p, "out", t))


def finetune_forward(params, ldr, hdr, table, thr=0.12):
    """finetune_real_dataset.py:144-172 in float64 with autograd."""
    c_pred = torch.clamp(deq_forward(params["deq"], ldr), 0, 1)
    invcrf = lin_forward(params["lin"], c_pred, table, True)
    b_pred = apply_rf(c_pred, invcrf)
    hal = hal_forward(params["hal"], b_pred, True)
    a_pred = b_pred + alpha_mask(b_pred, thr) * hal.flip(-1)
    r = ref_forward(params["ref"], torch.cat([a_pred, b_pred, c_pred], -1))
    r = r / (1e-6 + r.mean(dim=(1, 2, 3), keepdim=True)) * 0.5
    loss = (logc(r) - logc(hdr)).abs()
    return dict(loss=loss, C_pred=c_pred, B_pred=b_pred, A_pred=a_pred, refinement_output=r)

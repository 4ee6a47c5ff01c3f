"""NumPy restatement of the five networks and the step-closure arithmetic of the
SingleHDR hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Parameters are plain dicts  name -> ndarray.  Kernels are HWIO, as in Keras.
Names follow the reference's attribute names (e.g. "d2.conv1.kernel",
"crf_feature_net.res1.norm2.gamma") and the *order* of each `*_spec()` list
is Keras' `model.weights` order restricted to what the reference creates:
attribute-creation order, kernel before bias, gamma, beta, moving_mean,
moving_variance.
"""
import numpy as np

from . import ops

# --------------------------------------------------------------------------
# parameter specs
# --------------------------------------------------------------------------
def _conv(name, k, cin, cout, bias=True):
    s = [(name + ".kernel", (k, k, cin, cout), "kernel")]
    if bias:
        s.append((name + ".bias", (cout,), "bias"))
    return s


def _bn(name, c):
    return [(name + ".gamma", (c,), "gamma"), (name + ".beta", (c,), "beta"),
            (name + ".moving_mean", (c,), "moving_mean"),
            (name + ".moving_variance", (c,), "moving_variance")]


def unet_spec(cin, bottleneck):
    """dequantization_net.py:31-46 (cin=3, bottleneck=256) and
    refinement_net.py:31-47 (cin=9, bottleneck=128)."""
    s = _conv("conv1", 7, cin, 16) + _conv("conv2", 7, 16, 16)
    s += _conv("d2.conv1", 5, 16, 32) + _conv("d2.conv2", 5, 32, 32)
    s += _conv("d3.conv1", 3, 32, 64) + _conv("d3.conv2", 3, 64, 64)
    s += _conv("d4.conv1", 3, 64, 128) + _conv("d4.conv2", 3, 128, 128)
    s += _conv("enc.conv1", 3, 128, bottleneck) + _conv("enc.conv2", 3, bottleneck, bottleneck)
    s += _conv("u4.conv1", 3, bottleneck, 128) + _conv("u4.conv2", 3, 256, 128)
    s += _conv("u3.conv1", 3, 128, 64) + _conv("u3.conv2", 3, 128, 64)
    s += _conv("u2.conv1", 3, 64, 32) + _conv("u2.conv2", 3, 64, 32)
    s += _conv("u1.conv1", 3, 32, 16) + _conv("u1.conv2", 3, 32, 16)
    s += _conv("out", 3, 16, 3)
    return s


def deq_spec():
    return unet_spec(3, 256)


def ref_spec():
    return unet_spec(9, 128)


def _res1(name, cin, b1, b2):
    """linearization_net.py:6-26 (resBlock_type1)."""
    s = _conv(name + ".conv1", 1, cin, b1, False) + _bn(name + ".norm1", b1)
    s += _conv(name + ".conv2", 1, cin, b2[0], False) + _bn(name + ".norm2", b2[0])
    s += _conv(name + ".conv3", 3, b2[0], b2[1], False) + _bn(name + ".norm3", b2[1])
    s += _conv(name + ".conv4", 1, b2[1], b2[2], False) + _bn(name + ".norm4", b2[2])
    return s


def _res2(name, cin, f):
    """linearization_net.py:50-65 (resBlock_type2)."""
    s = _conv(name + ".conv1", 1, cin, f[0], False) + _bn(name + ".norm1", f[0])
    s += _conv(name + ".conv2", 3, f[0], f[1], False) + _bn(name + ".norm2", f[1])
    s += _conv(name + ".conv3", 1, f[1], f[2], False) + _bn(name + ".norm3", f[2])
    return s


def lin_spec():
    """linearization_net.py:85-101 (crfFeatureNet) + :185 (Dense 11)."""
    p = "crf_feature_net."
    s = _conv(p + "conv1", 7, 93, 64) + _bn(p + "norm1", 64)
    s += _res1(p + "res1", 64, 256, [64, 64, 256])
    s += _res2(p + "res2", 256, [64, 64, 256])
    s += _res2(p + "res3", 256, [64, 64, 256])
    s += _res1(p + "res4", 256, 512, [128, 128, 512])
    s += _res2(p + "res5", 512, [128, 128, 512])
    s += [("ae_invcrf_decode_net.fc.kernel", (512, 11), "kernel"),
          ("ae_invcrf_decode_net.fc.bias", (11,), "bias")]
    return s


def hal_spec():
    """hallucination_net.py:109-144.  `up.conv2` is constructed but never
    called in the reference (hallucination_net.py:83), so it owns no weights."""
    s = _conv("d1.conv1", 3, 3, 64) + _conv("d1.conv2", 3, 64, 64)
    s += _conv("d2.conv1", 3, 64, 128) + _conv("d2.conv2", 3, 128, 128)
    s += _conv("d3.conv1", 3, 128, 256) + _conv("d3.conv2", 3, 256, 256) + _conv("d3.conv3", 3, 256, 256)
    s += _conv("d4.conv1", 3, 256, 512) + _conv("d4.conv2", 3, 512, 512) + _conv("d4.conv3", 3, 512, 512)
    s += _conv("d5.conv1", 3, 512, 512) + _conv("d5.conv2", 3, 512, 512) + _conv("d5.conv3", 3, 512, 512)
    s += _conv("conv1", 3, 512, 512) + _bn("norm1", 512)
    for name, cin, cout in (("5", 512, 512), ("4", 512, 512), ("3", 512, 256),
                            ("2", 256, 128), ("1", 128, 64)):
        s += _conv("u%s.conv1" % name, 3, cin, cout) + _bn("u%s.norm1" % name, cout)
        s += _conv("s%s.conv1" % name, 1, 2 * cout, cout)
    s += _conv("conv2", 1, 64, 3) + _bn("norm2", 3)
    s += _conv("s0.conv1", 1, 6, 3)
    return s


def vgg_spec():
    """vgg16.py:69-83: conv1_1 .. conv3_3, constants from vgg16.npy."""
    s = []
    for name, cin, cout in (("conv1_1", 3, 64), ("conv1_2", 64, 64), ("conv2_1", 64, 128),
                            ("conv2_2", 128, 128), ("conv3_1", 128, 256),
                            ("conv3_2", 256, 256), ("conv3_3", 256, 256)):
        s += _conv(name, 3, cin, cout)
    return s


def trainable(spec):
    return [e for e in spec if e[2] in ("kernel", "bias", "gamma", "beta")]


def count_trainable(spec):
    return int(sum(int(np.prod(shape)) for _, shape, _ in trainable(spec)))


def init_params(spec, seed, randomize_bn=True, dtype=np.float32):
    """Keras default initialisers (glorot-uniform kernels, zero bias, BN
    gamma=1 beta=0 mean=0 var=1).  With randomize_bn the BN tensors and the
    biases are drawn non-trivially so that parity tests exercise them
    (mu ~ N(0,0.1), sigma^2 ~ U(0.5,1.5))."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape, role in spec:
        if role == "kernel":
            if len(shape) == 4:
                fan_in = shape[0] * shape[1] * shape[2]
                fan_out = shape[0] * shape[1] * shape[3]
            else:
                fan_in, fan_out = shape
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            v = rng.uniform(-lim, lim, size=shape)
        elif role == "bias":
            v = rng.normal(0, 0.05, size=shape) if randomize_bn else np.zeros(shape)
        elif role == "gamma":
            v = rng.uniform(0.5, 1.5, size=shape) if randomize_bn else np.ones(shape)
        elif role == "beta":
            v = rng.normal(0, 0.1, size=shape) if randomize_bn else np.zeros(shape)
        elif role == "moving_mean":
            v = rng.normal(0, 0.1, size=shape) if randomize_bn else np.zeros(shape)
        elif role == "moving_variance":
            v = rng.uniform(0.5, 1.5, size=shape) if randomize_bn else np.ones(shape)
        else:
            raise ValueError(role)
        p[name] = np.asarray(v, dtype=dtype)
    return p


# --------------------------------------------------------------------------
# forward passes
# --------------------------------------------------------------------------
def _c(p, name, x, stride=1):
    return ops.conv2d(x, p[name + ".kernel"], p.get(name + ".bias"), stride)


def _bnorm(p, name, x, training, stats=None):
    if training:
        y, m, v = ops.batch_norm_train(x, p[name + ".gamma"], p[name + ".beta"])
        if stats is not None:
            stats[name] = (m, v)
        return y
    return ops.batch_norm_infer(x, p[name + ".gamma"], p[name + ".beta"],
                                p[name + ".moving_mean"], p[name + ".moving_variance"])


def unet_forward(p, x, head):
    """dequantization_net.py:48-65 (head='deq') / refinement_net.py:49-66 (head='ref')."""
    lr = ops.leaky_relu

    def down(name, t):  # dequantization_net.py:11-15
        t = ops.avg_pool2(t)
        t = lr(_c(p, name + ".conv1", t))
        return lr(_c(p, name + ".conv2", t))

    def up(name, t, skip):  # dequantization_net.py:24-29
        t = ops.resize_bilinear_2x(t)
        t = lr(_c(p, name + ".conv1", t))
        return lr(_c(p, name + ".conv2", np.concatenate([t, skip], axis=-1)))

    t = lr(_c(p, "conv1", x))
    s1 = lr(_c(p, "conv2", t))
    s2 = down("d2", s1)
    s3 = down("d3", s2)
    s4 = down("d4", s3)
    t = down("enc", s4)
    t = up("u4", t, s4)
    t = up("u3", t, s3)
    t = up("u2", t, s2)
    t = up("u1", t, s1)
    t = _c(p, "out", t)
    if head == "deq":
        return x + np.tanh(t)
    return ops.relu(x[..., 0:3] + t)


def deq_forward(p, x):
    return unet_forward(p, x, "deq")


def ref_forward(p, x):
    return unet_forward(p, x, "ref")


def crf_feature_net_forward(p, x, training=False, stats=None):
    """linearization_net.py:103-118."""
    q = "crf_feature_net."

    def res1(name, t, stride):  # linearization_net.py:28-48
        n1 = _bnorm(p, name + ".norm1", _c(p, name + ".conv1", t, stride), training, stats)
        a2 = ops.relu(_bnorm(p, name + ".norm2", _c(p, name + ".conv2", t, stride), training, stats))
        a3 = ops.relu(_bnorm(p, name + ".norm3", _c(p, name + ".conv3", a2), training, stats))
        n4 = _bnorm(p, name + ".norm4", _c(p, name + ".conv4", a3), training, stats)
        return ops.relu(n1 + n4)

    def res2(name, t):  # linearization_net.py:67-83
        a1 = ops.relu(_bnorm(p, name + ".norm1", _c(p, name + ".conv1", t), training, stats))
        a2 = ops.relu(_bnorm(p, name + ".norm2", _c(p, name + ".conv2", a1), training, stats))
        n3 = _bnorm(p, name + ".norm3", _c(p, name + ".conv3", a2), training, stats)
        return ops.relu(t + n3)

    t = _c(p, q + "conv1", x, 2)
    t = ops.relu(_bnorm(p, q + "norm1", t, training, stats))
    t = ops.max_pool(t, 3, 2)
    t = res1(q + "res1", t, 1)
    t = res2(q + "res2", t)
    t = res2(q + "res3", t)
    t = res1(q + "res4", t, 2)
    t = res2(q + "res5", t)
    return ops.global_avg_pool(t)


def lin_forward(p, img, table, training=False, stats=None):
    """linearization_net.py:310-334.  `table` is float [1024,12]: g0 | hinv(1..11)."""
    feat = crf_feature_net_forward(p, ops.lin_frontend(img), training, stats)
    w = ops.dense(feat, p["ae_invcrf_decode_net.fc.kernel"], p["ae_invcrf_decode_net.fc.bias"])
    invcrf = ops.invcrf_pca_decode(w, table[:, 0], table[:, 1:12])
    return ops.increase(invcrf)


def hal_forward(p, x, training=False, stats=None):
    """hallucination_net.py:146-190."""
    bgr = ops.vgg_preprocess(x)

    def down(name, t, nconv):  # hallucination_net.py:51-75
        for i in range(1, nconv + 1):
            t = ops.relu(_c(p, "%s.conv%d" % (name, i), t))
        return ops.max_pool(t, 2, 2), t

    def up(name, t):  # hallucination_net.py:85-91
        t = ops.resize_bilinear_2x(t)
        t = ops.relu(_c(p, name + ".conv1", t))
        return ops.relu(_bnorm(p, name + ".norm1", t, training, stats))

    def skip(name, t, sk):  # hallucination_net.py:99-107
        sk = sk * np.asarray(1.0 / 255, dtype=sk.dtype)
        return _c(p, name + ".conv1", np.concatenate([t, sk], axis=-1))

    t, d1 = down("d1", bgr, 2)
    t, d2 = down("d2", t, 2)
    t, d3 = down("d3", t, 3)
    t, d4 = down("d4", t, 3)
    t, d5 = down("d5", t, 3)
    t = ops.relu(_bnorm(p, "norm1", _c(p, "conv1", t), training, stats))
    t = skip("s5", up("u5", t), d5)
    t = skip("s4", up("u4", t), d4)
    t = skip("s3", up("u3", t), d3)
    t = skip("s2", up("u2", t), d2)
    t = skip("s1", up("u1", t), d1)
    t = ops.relu(_bnorm(p, "norm2", _c(p, "conv2", t), training, stats))
    return ops.relu(skip("s0", t, bgr))


def vgg_forward(p, rgb):
    """vgg16.py:95-133: returns (pool1, pool2, pool3)."""
    t = ops.vgg_preprocess(rgb)
    t = ops.relu(_c(p, "conv1_1", t))
    t = ops.relu(_c(p, "conv1_2", t))
    p1 = ops.max_pool(t, 2, 2)
    t = ops.relu(_c(p, "conv2_1", p1))
    t = ops.relu(_c(p, "conv2_2", t))
    p2 = ops.max_pool(t, 2, 2)
    t = ops.relu(_c(p, "conv3_1", p2))
    t = ops.relu(_c(p, "conv3_2", t))
    t = ops.relu(_c(p, "conv3_3", t))
    p3 = ops.max_pool(t, 2, 2)
    return p1, p2, p3


# --------------------------------------------------------------------------
# step closures
# --------------------------------------------------------------------------
def inference(params, ldr, table, with_refinement=True, thr=0.12):
    """test_real_refinement.py:86-110.  `params` = {"deq":..., "lin":..., "hal":..., "ref":...}."""
    out = {}
    c_pred = np.clip(deq_forward(params["deq"], ldr), 0, 1)
    invcrf = lin_forward(params["lin"], c_pred, table)
    b_pred = ops.apply_rf(c_pred, invcrf)
    hal = hal_forward(params["hal"], b_pred)
    a_pred = ops.alpha_blend(b_pred, hal, thr)
    out.update(C_pred=c_pred, invcrf=invcrf, B_pred=b_pred, hal=hal, A_pred=a_pred)
    if with_refinement:
        out["hdr"] = ref_forward(params["ref"], np.concatenate([a_pred, b_pred, c_pred], axis=-1))
    return out


def joint_losses(params, vgg_params, batch, invcrf_gt, table, thr=0.12):
    """joint_training.py:137-183: the per-term losses of the joint step
    (training-mode BN).  `batch` = (ldr, jpeg, clipped_hdr_t, hdr_t, loss_mask)."""
    ldr, jpeg, clipped, hdr_t, mask = batch
    alpha = ops.alpha_mask(clipped, thr)
    c_pred = np.clip(deq_forward(params["deq"], jpeg), 0, 1)
    loss_deq = ops.l2_loss_with_mask(c_pred, ldr) * mask
    pred_invcrf = lin_forward(params["lin"], ldr, table, training=True)
    b_pred = ops.apply_rf(ldr, pred_invcrf)
    # joint_training.py:158-160: crf_loss keeps its [b,1] shape, so `10 * l2loss_lin [b,1,1,1] + crf_loss [b,1]` BROADCASTS to
    # [b,1,b,1] (element [i,0,j,0] = 10 * l2_i + crf_j) and so does everything added to it.  tape.gradient differentiates the
    # sum of all b*b elements: b * sum_i (per-sample terms)_i + (sum_i mask_i) * sum_j crf_j -- every per-sample term is
    # weighted by the batch size and a masked sample still receives crf gradient.  Restated as the reference computes it.
    crf_loss = ((pred_invcrf - invcrf_gt) ** 2).mean(axis=1, keepdims=True)
    loss_lin = (10.0 * ops.l2_loss_with_mask(b_pred, clipped) + crf_loss) * mask
    hal = hal_forward(params["hal"], clipped, training=True)
    a_pred = clipped + alpha * ops.reverse_channels(hal)
    ya = ops.log_compress(a_pred)
    yh = ops.log_compress(hdr_t)
    perc = 0
    for fa, fb in zip(vgg_forward(vgg_params, ya), vgg_forward(vgg_params, yh)):
        perc = perc + np.abs(fa - fb).mean(axis=(1, 2, 3), keepdims=True)
    l1 = ops.l1_loss_per_sample(ya, yh)
    tv = ops.tv_loss(ya)
    loss_hal = (l1 + 0.001 * perc + 0.1 * tv) * mask
    return dict(loss_deq=loss_deq, loss_lin=loss_lin, loss_hal=loss_hal, crf_loss=crf_loss,
                total=loss_deq + loss_lin + loss_hal, C_pred=c_pred, B_pred=b_pred, A_pred=a_pred)


def finetune_forward(params, ldr, hdr, table, thr=0.12):
    """finetune_real_dataset.py:144-172 (chained, training-mode BN; `_hal(pred)` read as `_hal(B_pred)`).
    Returns the un-reduced loss tensor and the intermediates."""
    c_pred = np.clip(deq_forward(params["deq"], ldr), 0, 1)
    invcrf = lin_forward(params["lin"], c_pred, table, training=True)
    b_pred = ops.apply_rf(c_pred, invcrf)
    hal = hal_forward(params["hal"], b_pred, training=True)
    a_pred = ops.alpha_blend(b_pred, hal, thr)
    r = ref_forward(params["ref"], np.concatenate([a_pred, b_pred, c_pred], axis=-1))
    r = r / (1e-6 + r.mean(axis=(1, 2, 3), keepdims=True)) * 0.5
    loss = np.abs(ops.log_compress(r) - ops.log_compress(hdr))
    return dict(loss=loss, C_pred=c_pred, B_pred=b_pred, A_pred=a_pred, refinement_output=r)

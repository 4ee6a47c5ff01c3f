"""NumPy restatement of the TF ops on the SingleHDR hot path (TEST INFRASTRUCTURE).

All tensors are NHWC; every function computes in the dtype of its first
argument (use float64 inputs for the "truth" oracle, float32 for the
bit-level histogram checks and the CPU baseline timing).

TF semantics encoded here follow SURVEY.md section 8c's list; each function cites the
reference call site it restates.
"""
import numpy as np

VGG_MEAN = (103.939, 116.779, 123.68)  # hallucination_net.py:110, vgg16.py:57


# --------------------------------------------------------------------------
# padding rule
# --------------------------------------------------------------------------
def same_pad(in_size, k, stride):
    """TF 'SAME' rule: out = ceil(in/s); total = max((out-1)*s + k - in, 0);
    before = total // 2 (the extra cell goes to the bottom/right)."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    before = total // 2
    return out, before, total - before


# --------------------------------------------------------------------------
# convolution / dense
# --------------------------------------------------------------------------
def conv2d(x, w, bias=None, stride=1, padding="SAME"):
    """Cross-correlation, NHWC x HWIO (tf.keras.layers.Conv2D / tf.nn.conv2d;
    e.g. dequantization_net.py:8-9, hallucination_net.py:47-48, vgg16.py:33).

    Implemented as one [pixels, Cin] @ [Cin, Cout] product per filter tap so
    that no im2col buffer is materialised."""
    n, h, wd, cin = x.shape
    kh, kw, cin_w, cout = w.shape
    assert cin == cin_w, (x.shape, w.shape)
    if padding == "SAME":
        ho, pt, pb = same_pad(h, kh, stride)
        wo, pl, pr = same_pad(wd, kw, stride)
    else:
        ho = (h - kh) // stride + 1
        wo = (wd - kw) // stride + 1
        pt = pb = pl = pr = 0
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    y = np.zeros((n, ho, wo, cout), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            win = xp[:, i:i + (ho - 1) * stride + 1:stride,
                     j:j + (wo - 1) * stride + 1:stride, :]
            y += (win.reshape(-1, cin) @ w[i, j].astype(x.dtype)).reshape(n, ho, wo, cout)
    if bias is not None:
        y = y + bias.astype(x.dtype)
    return y


def dense(x, w, b):
    """tf.keras.layers.Dense (linearization_net.py:185,192)."""
    return x @ w.astype(x.dtype) + b.astype(x.dtype)


# --------------------------------------------------------------------------
# activations / normalisation
# --------------------------------------------------------------------------
def relu(x):
    return np.maximum(x, 0)


def leaky_relu(x, alpha=0.1):
    """tf.nn.leaky_relu(x, 0.1) (dequantization_net.py:13-14)."""
    return np.where(x >= 0, x, x * np.asarray(alpha, dtype=x.dtype))


def batch_norm_infer(x, gamma, beta, mean, var, eps=1e-3):
    """Keras BatchNormalization, inference mode, eps 1e-3 (Keras default;
    linearization_net.py:13, hallucination_net.py:82)."""
    d = x.dtype
    inv = gamma.astype(d) / np.sqrt(var.astype(d) + np.asarray(eps, d))
    return x * inv + (beta.astype(d) - mean.astype(d) * inv)


def batch_norm_train(x, gamma, beta, eps=1e-3):
    """Training mode: normalise with the biased batch variance over (N,H,W).
    Returns (y, batch_mean, biased_batch_var)."""
    d = x.dtype
    mean = x.mean(axis=(0, 1, 2))
    var = ((x - mean) ** 2).mean(axis=(0, 1, 2))
    y = (x - mean) / np.sqrt(var + np.asarray(eps, d)) * gamma.astype(d) + beta.astype(d)
    return y, mean, var


# --------------------------------------------------------------------------
# pooling / resize
# --------------------------------------------------------------------------
def avg_pool2(x):
    """AveragePooling2D((2,2), strides=2), VALID (dequantization_net.py:10)."""
    n, h, w, c = x.shape
    ho, wo = h // 2, w // 2
    v = x[:, :ho * 2, :wo * 2, :].reshape(n, ho, 2, wo, 2, c)
    return v.mean(axis=(2, 4)).astype(x.dtype)


def max_pool(x, k, stride):
    """MaxPool2D(k, stride, SAME); padded cells never win
    (hallucination_net.py:49, linearization_net.py:94, vgg16.py:54)."""
    n, h, w, c = x.shape
    ho, pt, pb = same_pad(h, k, stride)
    wo, pl, pr = same_pad(w, k, stride)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
    y = np.full((n, ho, wo, c), -np.inf, dtype=x.dtype)
    for i in range(k):
        for j in range(k):
            y = np.maximum(y, xp[:, i:i + (ho - 1) * stride + 1:stride,
                                 j:j + (wo - 1) * stride + 1:stride, :])
    return y


def resize_bilinear_2x(x):
    """tf.image.resize(x, 2*shape, BILINEAR) in TF2: half-pixel centres, no
    antialias (dequantization_net.py:25, hallucination_net.py:86).

    src = (dst + 0.5)/2 - 0.5; lower = max(floor(src), 0); upper =
    min(ceil(src), in-1); lerp = src - floor(src)."""
    def axis_weights(n_in):
        dst = np.arange(2 * n_in, dtype=np.float64)
        src = (dst + 0.5) * 0.5 - 0.5
        f = np.floor(src)
        lo = np.maximum(f, 0).astype(np.int64)
        hi = np.minimum(np.ceil(src), n_in - 1).astype(np.int64)
        return lo, hi, (src - f)

    n, h, w, c = x.shape
    d = x.dtype
    ylo, yhi, yl = axis_weights(h)
    xlo, xhi, xl = axis_weights(w)
    xl = xl.astype(d)[None, None, :, None]
    yl = yl.astype(d)[None, :, None, None]
    top = x[:, ylo][:, :, xlo] + (x[:, ylo][:, :, xhi] - x[:, ylo][:, :, xlo]) * xl
    bot = x[:, yhi][:, :, xlo] + (x[:, yhi][:, :, xhi] - x[:, yhi][:, :, xlo]) * xl
    return top + (bot - top) * yl


def global_avg_pool(x):
    """tf.reduce_mean(x, [1, 2]) (linearization_net.py:118)."""
    return x.mean(axis=(1, 2))


# --------------------------------------------------------------------------
# Linearization-Net front end
# --------------------------------------------------------------------------
def sobel_edges(x):
    """tf.image.sobel_edges + reshape to 6 channels (linearization_net.py:312-314):
    REFLECT pad 1, dy kernel [[-1,-2,-1],[0,0,0],[1,2,1]], dx its transpose;
    output channel = c*2 + {0: dy, 1: dx}."""
    n, h, w, c = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)), mode="reflect")
    ky = np.array([[-1, -2, -1], [0, 0, 0], [1, 2, 1]], dtype=x.dtype)
    kx = ky.T
    dy = np.zeros_like(x)
    dx = np.zeros_like(x)
    for i in range(3):
        for j in range(3):
            win = xp[:, i:i + h, j:j + w, :]
            if ky[i, j] != 0:
                dy = dy + ky[i, j] * win
            if kx[i, j] != 0:
                dx = dx + kx[i, j] * win
    out = np.stack([dy, dx], axis=-1)  # [n,h,w,c,2]
    return out.reshape(n, h, w, c * 2)


def histogram_layer(img, max_bin):
    """linearization_net.py:336-350.  For i = 1..B:
        d = |img - (2i-1)/(2B)|;  h = d < 1/B ? 1 - d*B : 0
    concatenated over i => channel order [bin1.RGB, bin2.RGB, ...].

    In float32 the centre is formed as fp32(2i-1)/fp32(2B) (tf.divide of two
    fp32 scalars) and the threshold as fp32(1/B) (a Python double cast to
    fp32), SURVEY.md section 8c item 11."""
    d = img.dtype
    thr = np.asarray(1.0 / max_bin, dtype=d)
    nb = np.asarray(max_bin, dtype=d)
    one = np.asarray(1.0, dtype=d)
    outs = []
    for i in range(1, max_bin + 1):
        centre = np.asarray(2.0 * i - 1.0, dtype=d) / np.asarray(2.0 * max_bin, dtype=d)
        dist = np.abs(img - centre)
        outs.append(np.where(dist < thr, one - dist * nb, np.asarray(0, dtype=d)))
    return np.concatenate(outs, axis=-1)


def lin_frontend(img):
    """linearization_net.py:322: concat[img, sobel(6), hist4, hist8, hist16] = 93 ch."""
    return np.concatenate([img, sobel_edges(img), histogram_layer(img, 4),
                           histogram_layer(img, 8), histogram_layer(img, 16)], axis=-1)


# --------------------------------------------------------------------------
# inverse-CRF head
# --------------------------------------------------------------------------
def invcrf_pca_decode(wts, g0, hinv):
    """linearization_net.py:231-253: invcrf = g0 + HINV[1024,11] @ w[b,11]."""
    d = wts.dtype
    return g0.astype(d)[None, :] + wts @ hinv.astype(d).T


def increase(rf):
    """linearization_net.py:368-392 (`_increase`)."""
    g = rf[:, 1:] - rf[:, :-1]
    min_g = g.min(axis=-1, keepdims=True)
    r = np.maximum(-min_g, 0)
    new_g = g + r
    new_g = new_g / new_g.sum(axis=-1, keepdims=True)
    new_rf = np.cumsum(new_g, axis=-1)
    return np.pad(new_rf, ((0, 0), (1, 0)))


def apply_rf(x, rf):
    """tf_utils.py:54-105 (apply_rf / interp_1d / sample_1d): per batch row,
    y = (k-1)*x; y0 = floor(y); y1 = y0+1; indices clipped to [0,k-1];
    out = (y1-y)*rf[y0] + (y-y0)*rf[y1]."""
    b = x.shape[0]
    k = rf.shape[1]
    d = x.dtype
    y = np.asarray(k - 1, dtype=d) * x.reshape(b, -1)
    y0 = np.floor(y)
    y1 = y0 + 1
    i0 = np.clip(y0.astype(np.int64), 0, k - 1)
    i1 = np.clip(y1.astype(np.int64), 0, k - 1)
    v0 = np.take_along_axis(rf.astype(d), i0, axis=1)
    v1 = np.take_along_axis(rf.astype(d), i1, axis=1)
    out = (y1 - y) * v0 + (y - y0) * v1
    return out.reshape(x.shape)


# --------------------------------------------------------------------------
# colour / glue arithmetic of the step closures
# --------------------------------------------------------------------------
def reverse_channels(x):
    """tf_utils.bgr2rgb == tf_utils.rgb2bgr (tf_utils.py:5-13): channel reversal."""
    return x[..., ::-1]


def vgg_preprocess(x):
    """x*255, RGB->BGR, subtract VGG_MEAN (hallucination_net.py:149-153, vgg16.py:101-109)."""
    d = x.dtype
    s = x * np.asarray(255.0, d)
    return np.stack([s[..., 2] - np.asarray(VGG_MEAN[0], d),
                     s[..., 1] - np.asarray(VGG_MEAN[1], d),
                     s[..., 0] - np.asarray(VGG_MEAN[2], d)], axis=-1)


def alpha_mask(x, thr=0.12):
    """test_real_refinement.py:98-101 / joint_training.py:141-145:
    alpha = min(1, max(0, max_c(x) - 1 + thr) / thr), tiled to 3 channels."""
    d = x.dtype
    a = x.max(axis=3, keepdims=True)
    a = np.minimum(np.asarray(1.0, d),
                   np.maximum(np.asarray(0.0, d), a - np.asarray(1.0, d) + np.asarray(thr, d)) / np.asarray(thr, d))
    return np.tile(a, (1, 1, 1, 3))


def alpha_blend(b_pred, hal_bgr, thr=0.12):
    """A = B + alpha * swapRB(hal) (test_real_refinement.py:103-105)."""
    return b_pred + alpha_mask(b_pred, thr) * reverse_channels(hal_bgr)


def log_compress(x):
    """log(1 + 10x) / log(11) (joint_training.py:166,173)."""
    d = x.dtype
    return np.log(np.asarray(1.0, d) + np.asarray(10.0, d) * x) / np.log(np.asarray(11.0, d))


def l2_loss_with_mask(a, b):
    """tf_utils.get_l2_loss_with_mask (tf_utils.py:110-111): per-sample MSE [b,1,1,1]."""
    return ((a - b) ** 2).mean(axis=(1, 2, 3), keepdims=True)


def l1_loss_per_sample(a, b):
    return np.abs(a - b).mean(axis=(1, 2, 3), keepdims=True)


def tv_loss(y):
    """joint_training.py:175-179: SYMMETRIC pad by one at the bottom / right
    (duplicates the edge sample, contributing a zero difference) then the
    batch-global mean of |forward differences|."""
    py = np.pad(y, ((0, 0), (0, 1), (0, 0), (0, 0)), mode="symmetric")
    px = np.pad(y, ((0, 0), (0, 0), (0, 1), (0, 0)), mode="symmetric")
    return np.abs(py[:, 1:] - py[:, :-1]).mean() + np.abs(px[:, :, 1:] - px[:, :, :-1]).mean()

"""Tensor utilities on the hot path -- drop-in for the hot-path half of the reference `tf_utils`.

tf_utils.py:5-13 (bgr2rgb / rgb2bgr), :19-27 (get_tensor_shape), :54-105
(apply_rf / interp_1d / sample_1d).  Run plumbing (TensorBoard writers,
tf.train.Checkpoint) is out of scope.
"""
try:
    from . import _ops as K
except ImportError:
    import _ops as K


def rgb2bgr(rgb):
    return K.reverse3(rgb)


def bgr2rgb(bgr):
    return K.reverse3(bgr)


def get_tensor_shape(x):
    return list(x.shape)


def apply_rf(x, rf):
    """x [b, s...] in [0,1], rf [b, k]: per-row LUT with linear interpolation."""
    return K.apply_rf(x, rf)

"""Step closures of the hot path, restated on the HIP ops.

`inference` follows test_real_refinement.py:86-110 of the reference
(deq -> clip -> lin -> apply_rf -> alpha -> hal -> blend -> ref).
"""
try:
    from . import _ops as K
    from . import tf_utils
except ImportError:
    import _ops as K
    import tf_utils

THRESHOLD = 0.12  # test_real_refinement.py:28


class Inference:
    """Callable equivalent of the reference's `inference(ldr)` tf.function."""

    def __init__(self, deq, lin, hal, ref=None, threshold=THRESHOLD):
        self._deq, self._lin, self._hal, self._ref = deq, lin, hal, ref
        self.threshold = threshold

    def __call__(self, ldr, return_intermediates=False):
        pred_deq = self._deq(ldr, training=False)
        C_pred = K.clip(pred_deq, 0.0, 1.0)
        pred_invcrf = self._lin(C_pred, training=False)
        B_pred = tf_utils.apply_rf(C_pred, pred_invcrf)
        bgr_hal_res = self._hal(B_pred, training=False)
        # alpha = clamp((max_c B - 1 + thr)/thr); A = B + alpha * rgb2bgr(hal)   (:98-105), one kernel
        A_pred = K.alpha_blend(B_pred, bgr_hal_res, self.threshold)
        out = A_pred
        if self._ref is not None:
            # tf.concat([A,B,C],-1) (:108), zero-padded to 12 channels for the MFMA tile
            out = self._ref(K.pack3([A_pred, B_pred, C_pred], 12), training=False)
        if return_intermediates:
            return dict(C_pred=C_pred, invcrf=pred_invcrf, B_pred=B_pred, hal=bgr_hal_res, A_pred=A_pred,
                        hdr=out if self._ref is not None else None)
        return out

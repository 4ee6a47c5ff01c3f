"""`_preprocessing(hdr, crf, t)` of joint_training.py:26-69 on the device.

The reference builds every training batch from an HDR crop with tf ops plus a per-image libjpeg encode/decode on the
host (tf.image.adjust_jpeg_quality).  Here the whole chain is three libshdr launches -- exposure + noise + clip,
the CRF (the same apply_rf kernel the networks use), and the JPEG round trip + loss mask -- so the simulator keeps up
with the accelerated train step.  Noise comes from a counter-based Philox stream keyed by `seed` (TF's stateful
generators cannot be reproduced bit for bit; the DISTRIBUTIONS are the reference's).
"""
import torch

try:
    from . import _ops as K
    from . import tf_utils
except ImportError:
    import _ops as K
    import tf_utils


def jpeg_qualities(batch_size):
    """joint_training.py:48: quality of sample i = int(round(i / (BATCH_SIZE - 1) * 10 + 90)): 90 ... 100 over the batch"""
    if batch_size == 1:
        return [90]
    return [int(round(float(i) / float(batch_size - 1) * 10.0 + 90.0)) for i in range(batch_size)]


class CameraPipeline:
    def __init__(self, seed=1):
        self.seed, self.calls = int(seed), 0
        self._q = {}

    def __call__(self, hdr, crf, t):
        """hdr [b,h,w,3] (linear radiance), crf [b,1024] (camera response), t [b] (exposure) ->
        [ldr, jpeg_img_float, clipped_hdr_t, hdr_t, loss_mask] exactly as `_preprocessing` returns them"""
        b = hdr.shape[0]
        q = self._q.get((b, hdr.device))
        if q is None:
            q = self._q[(b, hdr.device)] = torch.tensor(jpeg_qualities(b), dtype=torch.int32, device=hdr.device)
        with torch.no_grad():
            hdr_t, clipped = K.camera_expose(hdr, t, self.seed + (self.calls << 20))
            self.calls += 1
            ldr = tf_utils.apply_rf(clipped, crf)
            jpeg, mask = K.jpeg_round_trip(ldr, q)
        return [ldr, jpeg, clipped, hdr_t, mask]

"""singlehdr-tf2_amd: MI355X-native (gfx950) hot path of single-image HDR reconstruction.

Drop-in modules with the reference's Python call surface:
`dequantization_net`, `linearization_net`, `hallucination_net`,
`refinement_net` (each with class `model`), `vgg16.Vgg16`, `tf_utils`.
Put this directory on `sys.path` to use them under the reference's module
names, or import them from this package.  All arithmetic runs in libshdr.so
(hand-written HIP, C ABI in include/shdr.h); there is no CPU fallback.
"""
from . import _lib, _ops, _autograd, _layers  # noqa: F401
from . import dequantization_net, linearization_net, hallucination_net, refinement_net  # noqa: F401
from . import vgg16, tf_utils, pipeline, tf_checkpoint, hdr_io, camera, tfrecord  # noqa: F401

__all__ = ["dequantization_net", "linearization_net", "hallucination_net", "refinement_net",
           "vgg16", "tf_utils", "pipeline", "tf_checkpoint", "hdr_io", "camera", "tfrecord"]

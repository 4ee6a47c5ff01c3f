"""Tensor utilities on the hot path -- drop-in for the hot-path half of the reference `tf_utils`.

tf_utils.py:5-13 (bgr2rgb / rgb2bgr), :19-27 (get_tensor_shape), :54-105
(apply_rf / interp_1d / sample_1d), :110-111 (get_l2_loss_with_mask), :149-169 (checkpoint_initialization, on the
TensorFlow-free checkpoint reader / writer of tf_checkpoint.py).  TensorBoard writers are out of scope.
"""
import os

try:
    from . import _ops as K
    from . import tf_checkpoint
except ImportError:
    import _ops as K
    import tf_checkpoint


def rgb2bgr(rgb):
    return K.reverse3(rgb)


def bgr2rgb(bgr):
    return K.reverse3(bgr)


def get_tensor_shape(x):
    return list(x.shape)


def apply_rf(x, rf):
    """x [b, s...] in [0,1], rf [b, k]: per-row LUT with linear interpolation."""
    return K.apply_rf(x, rf)


def get_l2_loss_with_mask(pred, gt):
    """tf.reduce_mean(tf.square(pred - gt), axis=[1, 2, 3], keepdims=True): per-sample MSE [b, 1, 1, 1]  (tf_utils.py:110-111)"""
    return K.diff_loss(pred, gt, 0).reshape(-1, 1, 1, 1)


class _Counter:
    """the `epoch = tf.Variable(0)` of a checkpoint: .assign_add(n) / int()"""

    def __init__(self, value=0):
        self.value = int(value)

    def assign_add(self, n):
        self.value += int(n)
        return self

    def assign(self, v):
        self.value = int(v)
        return self

    def numpy(self):
        return self.value

    def __int__(self):
        return self.value


class Checkpoint:
    """tf.train.Checkpoint(epoch=tf.Variable(0), lin=model, optimizer=optimizer) as tf_utils.py:156-159 builds it"""

    def __init__(self, model, optimizer=None):
        self.epoch, self.lin, self.optimizer = _Counter(0), model, optimizer

    def restore(self, path):
        info = tf_checkpoint.restore(self.lin, path, root="lin", optimizer=self.optimizer)
        if info["epoch"] is not None:
            self.epoch.assign(info["epoch"])
        return info


class CheckpointManager:
    """tf.train.CheckpointManager(ckpt, directory, max_to_keep=5): .latest_checkpoint, .save()"""

    def __init__(self, checkpoint, directory, max_to_keep=5):
        self.checkpoint, self.directory, self.max_to_keep = checkpoint, directory, max_to_keep
        latest = tf_checkpoint.latest_checkpoint(directory)
        self._counter = int(latest.rsplit("-", 1)[1]) if latest and latest.rsplit("-", 1)[-1].isdigit() else 0

    @property
    def latest_checkpoint(self):
        return tf_checkpoint.latest_checkpoint(self.directory)

    def save(self):
        self._counter += 1
        return tf_checkpoint.save(self.directory, self.checkpoint.lin, root="lin", epoch=int(self.checkpoint.epoch),
                                  optimizer=self.checkpoint.optimizer, save_counter=self._counter, max_to_keep=self.max_to_keep)


def checkpoint_initialization(model_name, pretrained_dirpath, model="model", optimizer="optimizer"):
    """tf_utils.py:149-169: make the directory, build (ckpt, ckpt_manager), restore the latest checkpoint if there is one.
    `optimizer`: a pipeline.KerasAdam over exactly this model's variables, or None."""
    os.makedirs(pretrained_dirpath, exist_ok=True)
    ckpt = Checkpoint(model, None if isinstance(optimizer, str) else optimizer)
    ckpt_manager = CheckpointManager(ckpt, pretrained_dirpath, max_to_keep=5)
    if ckpt_manager.latest_checkpoint:
        ckpt.restore(ckpt_manager.latest_checkpoint)
        print("Latest {} checkpoint has restored!!".format(model_name))
    return ckpt, ckpt_manager

"""HDR-Real TFRecord files without TensorFlow.

convert_to_tf_record.py:62-66 writes GZIP-compressed TFRecord files whose records are tf.train.Example protos with two
bytes features, `ref_HDR` and `ref_LDR`: the raw float32 bytes of a 256x256x3 patch each.  finetune_real_dataset.py:34-78
reads them back: decode_raw, reshape, HDR / (1e-6 + mean) * 0.5, LDR / 255, a random horizontal flip and a random rot90,
shuffle, batch 4.  Host work here is only the container (zlib, record framing with masked CRC-32C, the protobuf map);
the normalisation and the augmentation run on the device (`mean_norm`, `flip_rot90` kernels).

Record framing (tensorflow/core/lib/io/record_writer.cc): uint64 length | uint32 masked_crc32c(length) | data |
uint32 masked_crc32c(data), little endian.  Restated from the published format: no TF-written file has been read here.
"""
import glob
import gzip
import os
import struct

import numpy as np
import torch

try:
    from . import _ops as K
    from .tf_checkpoint import crc32c, mask_crc, _fields, _f_bytes
except ImportError:
    import _ops as K
    from tf_checkpoint import crc32c, mask_crc, _fields, _f_bytes

IMSHAPE = (256, 256, 3)        # finetune_real_dataset.py:28
BATCH_SIZE = 4                 # finetune_real_dataset.py:25


def _open(path, mode, compression):
    if compression in ("GZIP", "gzip"):
        return gzip.open(path, mode)
    if compression in (None, "", "NONE"):
        return open(path, mode)
    raise ValueError("compression_type %r is not supported (GZIP or none)" % compression)


def read_records(path, compression="GZIP", verify=True):
    """yield the payload of every record of one TFRecord file"""
    with _open(path, "rb", compression) as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) != 12:
                raise ValueError("%s: truncated record header" % path)
            length, lcrc = struct.unpack("<QI", head)
            if verify and mask_crc(crc32c(head[:8])) != lcrc:
                raise ValueError("%s: corrupt record length" % path)
            data = f.read(length)
            tail = f.read(4)
            if len(data) != length or len(tail) != 4:
                raise ValueError("%s: truncated record" % path)
            if verify and mask_crc(crc32c(data)) != struct.unpack("<I", tail)[0]:
                raise ValueError("%s: record fails its CRC-32C" % path)
            yield data


def write_records(path, records, compression="GZIP"):
    with _open(path, "wb", compression) as f:
        for data in records:
            head = struct.pack("<Q", len(data))
            f.write(head + struct.pack("<I", mask_crc(crc32c(head))) + data + struct.pack("<I", mask_crc(crc32c(data))))


def parse_example(buf):
    """tf.train.Example -> {name: [bytes, ...]} for bytes features (float / int64 lists come back as raw packed bytes)"""
    out = {}
    for fno, _, features in _fields(buf):                      # Example.features = 1
        if fno != 1:
            continue
        for f2, _, entry in _fields(features):                 # Features.feature = 1 (map entries)
            if f2 != 1:
                continue
            name, values = None, []
            for f3, _, v in _fields(entry):                    # key = 1, value = 2 (Feature)
                if f3 == 1:
                    name = v.decode()
                elif f3 == 2:
                    for f4, _, lst in _fields(v):              # bytes_list = 1 | float_list = 2 | int64_list = 3
                        values = [val for f5, _, val in _fields(lst) if f5 == 1]
            if name is not None:
                out[name] = values
    return out


def make_example(features):
    """{name: bytes} -> serialized tf.train.Example with one-element bytes features (convert_to_tf_record.py:9-10, 62-65)"""
    body = b""
    for name in sorted(features):                              # protobuf serialises map entries in key order deterministically
        feature = _f_bytes(1, _f_bytes(1, bytes(features[name])))          # Feature{ bytes_list{ value } }
        body += _f_bytes(1, _f_bytes(1, name.encode()) + _f_bytes(2, feature))
    return _f_bytes(1, body)


class HdrRealDataset:
    """`configureDataset(dirpath)` of finetune_real_dataset.py:63-78: iterate (ref_LDR, ref_HDR) batches on the device.

    shuffle_buffer=None reproduces the reference's full shuffle (buffer = number of FILES, :72, i.e. a small window over
    the record stream); augmentation draws come from `seed`."""

    def __init__(self, dirpath, batch_size=BATCH_SIZE, compression="GZIP", seed=0, augment=True, shuffle_buffer=None,
                 imshape=IMSHAPE, device=None):
        self.files = sorted(glob.glob(os.path.join(dirpath, "*.tfrecords")))
        if not self.files:
            raise FileNotFoundError("no *.tfrecords under %s" % dirpath)
        self.batch_size, self.compression, self.augment, self.imshape = batch_size, compression, augment, tuple(imshape)
        self.shuffle_buffer = len(self.files) if shuffle_buffer is None else shuffle_buffer
        self.rng = np.random.default_rng(seed)
        self.device = device or torch.device("cuda", torch.cuda.current_device())

    def _patches(self):
        n = int(np.prod(self.imshape))
        for path in self.files:
            for rec in read_records(path, self.compression):
                ex = parse_example(rec)
                hdr = np.frombuffer(ex["ref_HDR"][0], dtype="<f4")       # tf.io.decode_raw(..., tf.float32)   (:43-44)
                ldr = np.frombuffer(ex["ref_LDR"][0], dtype="<f4")
                if hdr.size != n or ldr.size != n:
                    raise ValueError("%s: patch of %d floats, expected %s" % (path, hdr.size, self.imshape))
                yield ldr.reshape(self.imshape), hdr.reshape(self.imshape)

    def _shuffled(self):
        buf = []
        for item in self._patches():                           # tf.data shuffle: a reservoir window of `buffer_size`
            if len(buf) < max(self.shuffle_buffer, 1):
                buf.append(item)
                continue
            k = int(self.rng.integers(len(buf)))
            out, buf[k] = buf[k], item
            yield out
        self.rng.shuffle(buf)
        yield from buf

    def __iter__(self):
        batch = []
        for item in self._shuffled():
            batch.append(item)
            if len(batch) == self.batch_size:
                yield self._to_device(batch)
                batch = []
        if batch:                                              # drop_remainder=False (:72)
            yield self._to_device(batch)

    def _to_device(self, batch):
        ldr = torch.from_numpy(np.stack([b[0] for b in batch])).to(self.device)
        hdr = torch.from_numpy(np.stack([b[1] for b in batch])).to(self.device)
        n = len(batch)
        u = self.rng.random((n, 2)).astype(np.float32) if self.augment else np.ones((n, 2), dtype=np.float32)
        flip = torch.from_numpy((u[:, 0] < 0.5).astype(np.int32)).to(self.device)              # :54-55
        rot = torch.from_numpy((u[:, 1] * 4 + 0.5).astype(np.int32) if self.augment else np.zeros(n, dtype=np.int32)).to(self.device)  # :58
        with torch.no_grad():
            hdr = K.mean_norm(hdr, 1e-6, 0.5)                                                  # :48
            return K.flip_rot90(ldr, flip, rot, 255.0), K.flip_rot90(hdr, flip, rot, 1.0)     # :49-60

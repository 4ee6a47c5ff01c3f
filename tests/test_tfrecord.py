"""TFRecord container + tf.train.Example restatement (convert_to_tf_record.py:62-66, finetune_real_dataset.py:34-46). CPU."""
import gzip
import importlib
import struct

import numpy as np
import pytest

pkg = importlib.import_module("singlehdr-tf2_amd")
T = pkg.tfrecord


def test_example_wire_bytes_and_parse():
    ex = T.make_example({"a": b"xy"})
    assert ex == bytes([0x0a, 0x0d, 0x0a, 0x0b, 0x0a, 0x01, 0x61, 0x12, 0x06, 0x0a, 0x04, 0x0a, 0x02, 0x78, 0x79])
    assert T.parse_example(ex) == {"a": [b"xy"]}
    both = T.parse_example(T.make_example({"ref_LDR": b"\x01\x02", "ref_HDR": b"\x03" * 300}))
    assert both == {"ref_HDR": [b"\x03" * 300], "ref_LDR": [b"\x01\x02"]}


@pytest.mark.parametrize("compression", ["GZIP", None])
def test_record_framing_round_trip(tmp_path, compression):
    recs = [b"", b"a", bytes(range(256)) * 40, b"tail"]
    path = str(tmp_path / "r.tfrecords")
    T.write_records(path, recs, compression)
    assert list(T.read_records(path, compression)) == recs
    raw = (gzip.open(path, "rb") if compression else open(path, "rb")).read()
    assert raw[:8] == struct.pack("<Q", 0) and len(raw) == sum(16 + len(r) for r in recs)
    # masked CRC of the 8 length bytes of an empty record, and of empty data (crc32c("") = 0 -> mask = 0xa282ead8)
    assert struct.unpack("<I", raw[12:16])[0] == 0xa282ead8
    bad = bytearray(raw)
    bad[16 + 17 + 12 + 5] ^= 1                              # a payload byte of the third record
    (gzip.open(path, "wb") if compression else open(path, "wb")).write(bytes(bad))
    with pytest.raises(ValueError):
        list(T.read_records(path, compression))
    assert len(list(T.read_records(path, compression, verify=False))) == 4
    if compression:
        with pytest.raises(ValueError, match="compression"):
            list(T.read_records(path, "ZLIB"))

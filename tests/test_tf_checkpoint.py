"""TensorFlow tensor-bundle / object-graph checkpoint reader and writer, CPU only.

TensorFlow is not installable here, so the format is pinned by published known answers (RFC 3720 CRC-32C vectors,
LevelDB's table magic, hand-assembled protobuf bytes) and by round trips through the module's own writer."""
import importlib
import os
import struct

import numpy as np
import pytest
import torch

pkg = importlib.import_module("singlehdr-tf2_amd")
C = pkg.tf_checkpoint


def test_crc32c_known_answers():
    assert C.crc32c(b"123456789") == 0xE3069283                       # the CRC catalogue's check value
    assert C.crc32c(bytes(32)) == 0x8A9136AA                          # RFC 3720 B.4
    assert C.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert C.crc32c(bytes(range(32))) == 0x46DD794E
    assert C.crc32c(b"6789", C.crc32c(b"12345")) == 0xE3069283        # incremental
    a = np.arange(1000, dtype=np.float32)
    assert C.crc32c_array(a) == C.crc32c(a.tobytes())
    # LevelDB crc32c::Mask: rotate right by 15, add 0xa282ead8
    assert C.mask_crc(0) == 0xa282ead8 and C.mask_crc(0x00008000) == (1 + 0xa282ead8)


def test_bundle_entry_wire_bytes():
    e = C.BundleEntry(C.DT_FLOAT, (3, 3, 4, 16), offset=300, size=2304, crc=0x01020304)
    want = bytes([0x08, 0x01,                                         # 1: dtype = DT_FLOAT
                  0x12, 0x10, 0x12, 0x02, 0x08, 0x03, 0x12, 0x02, 0x08, 0x03,   # 2: shape { dim{size:3} dim{size:3}
                  0x12, 0x02, 0x08, 0x04, 0x12, 0x02, 0x08, 0x10,               #             dim{size:4} dim{size:16} }
                  0x20, 0xAC, 0x02,                                   # 4: offset = 300
                  0x28, 0x80, 0x12,                                   # 5: size = 2304
                  0x35, 0x04, 0x03, 0x02, 0x01])                      # 6: crc32c fixed32
    assert e.serialize() == want
    back = C.BundleEntry.parse(want)
    assert (back.dtype, back.shape, back.offset, back.size, back.crc32c) == (1, (3, 3, 4, 16), 300, 2304, 0x01020304)


def test_table_round_trip_multi_block(tmp_path):
    items = [(b"", b"header")] + [(("net/layer%03d/kernel/.ATTRIBUTES/VARIABLE_VALUE" % i).encode(), os.urandom(20 + i % 7))
                                  for i in range(400)]
    items.sort()
    path = str(tmp_path / "t.index")
    C.write_table(path, items, block_size=512)
    raw = open(path, "rb").read()
    assert raw[-8:] == bytes([0x57, 0xfb, 0x80, 0x8b, 0x24, 0x75, 0x47, 0xdb])      # LevelDB kTableMagicNumber
    assert len(raw) > 20 * 512 // 2                                                    # really several blocks
    got = C.read_table(path)
    assert list(got.items()) == items
    # a flipped payload byte is caught by the block CRC
    bad = bytearray(raw)
    bad[100] ^= 0x40
    open(path, "wb").write(bytes(bad))
    with pytest.raises(ValueError, match="CRC"):
        C.read_table(path)
    with pytest.raises(ValueError, match="increasing"):
        C.write_table(path, [(b"b", b"1"), (b"a", b"2")])


def test_object_graph_round_trip():
    a, b, c = C.GraphNode(), C.GraphNode(), C.GraphNode()
    a.children = {"lin": 1, "epoch": 2}
    b.attributes = {"VARIABLE_VALUE": "lin/kernel/.ATTRIBUTES/VARIABLE_VALUE"}
    a.slots = [(1, "m", 2)]
    nodes = C.parse_object_graph(C.serialize_object_graph([a, b, c]))
    assert nodes[0].children == a.children and nodes[1].attributes == b.attributes and nodes[0].slots == [(1, "m", 2)]
    assert nodes[2].children == {} and len(nodes) == 3


@pytest.mark.parametrize("net", ["dequantization_net", "linearization_net", "hallucination_net", "refinement_net"])
def test_model_checkpoint_round_trip(tmp_path, net):
    torch.manual_seed(1)
    src = getattr(pkg, net).model(device=torch.device("cpu"))
    for _, t, _ in src.named_weights():                    # BN statistics and biases away from their defaults
        with torch.no_grad():
            t.add_(torch.randn(t.shape) * 0.1)
    prefix = C.save(str(tmp_path), src, epoch=7)
    assert os.path.basename(prefix) == "ckpt-1" and C.latest_checkpoint(str(tmp_path)) == prefix
    rd = C.BundleReader(prefix)
    keys = rd.keys()
    first = src.named_weights()[0][0].replace(".", "/")
    assert "lin/%s/.ATTRIBUTES/VARIABLE_VALUE" % first in keys and "_CHECKPOINTABLE_OBJECT_GRAPH" in keys
    assert "epoch/.ATTRIBUTES/VARIABLE_VALUE" in keys and "save_counter/.ATTRIBUTES/VARIABLE_VALUE" in keys
    rd.close()
    torch.manual_seed(2)
    dst = getattr(pkg, net).model(device=torch.device("cpu"))
    info = C.restore(dst, str(tmp_path))                   # directory -> latest checkpoint
    assert info["epoch"] == 7 and not info["missing"] and len(info["restored"]) == len(src.named_weights())
    for (n, a, _), (_, b, _) in zip(src.named_weights(), dst.named_weights()):
        assert torch.equal(a, b), n


def test_layer_with_weights_aliases_resolve(tmp_path):
    """Keras may spell a path `layer_with_weights-N` instead of the attribute name: both must resolve."""
    torch.manual_seed(3)
    src = pkg.linearization_net.model(device=torch.device("cpu"))
    prefix = C.save(str(tmp_path), src)
    rd = C.BundleReader(prefix)
    nodes = rd.object_graph()
    rd.close()
    for node in nodes:                                     # keep ONLY the alias spelling wherever one exists
        aliased = {nid for name, nid in node.children.items() if name.startswith("layer_with_weights-")}
        node.children = {k: v for k, v in node.children.items() if k.startswith("layer_with_weights-") or v not in aliased}
    assert "crf_feature_net" not in nodes[nodes[0].children["lin"]].children
    found, missing = {}, []
    C._walk(nodes, nodes[0].children["lin"], src, "", found, missing)
    assert not missing and len(found) == len(src.named_weights())
    assert found["crf_feature_net.res1.conv1.kernel"][0] == "lin/crf_feature_net/res1/conv1/kernel/.ATTRIBUTES/VARIABLE_VALUE"


def test_optimizer_slots_and_manager_state(tmp_path):
    torch.manual_seed(4)
    m = pkg.dequantization_net.model(device=torch.device("cpu"))
    fp = pkg.pipeline.FlatParams([m])
    opt = pkg.pipeline.KerasAdam(fp, 1e-5)
    fp.m.copy_(torch.randn(fp.numel))
    fp.v.copy_(torch.rand(fp.numel))
    opt.t = 1234
    for k in range(1, 8):
        C.save(str(tmp_path), m, optimizer=opt, save_counter=k, max_to_keep=5)
    files = sorted(os.listdir(str(tmp_path)))
    assert "ckpt-2.index" not in files and "ckpt-3.index" in files and "ckpt-7.data-00000-of-00001" in files
    assert C.latest_checkpoint(str(tmp_path)).endswith("ckpt-7")
    rd = C.BundleReader(C.latest_checkpoint(str(tmp_path)))
    assert "lin/conv1/kernel/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE" in rd.keys()
    assert int(rd.read("optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE").reshape(-1)[0]) == 1234
    rd.close()
    m2 = pkg.dequantization_net.model(device=torch.device("cpu"))
    fp2 = pkg.pipeline.FlatParams([m2])
    opt2 = pkg.pipeline.KerasAdam(fp2, 1e-5)
    C.restore(m2, str(tmp_path), optimizer=opt2)
    assert opt2.t == 1234
    for v, off in zip(fp.variables, fp.offsets):           # the alignment gaps of the flat buffers are not variables
        k = v.numel()
        assert torch.equal(fp2.m[off:off + k], fp.m[off:off + k]) and torch.equal(fp2.v[off:off + k], fp.v[off:off + k])
    assert torch.equal(fp2.flat, fp.flat)


def test_corrupt_tensor_and_wrong_root(tmp_path):
    m = pkg.refinement_net.model(device=torch.device("cpu"))
    prefix = C.save(str(tmp_path), m)
    with pytest.raises(KeyError, match="no object"):
        C.restore(m, prefix, root="model")
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    raw[len(raw) // 2] ^= 1
    open(data, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="CRC"):
        C.restore(m, prefix)
    C.restore(m, prefix, verify=False)                      # explicit opt-out still loads
    with pytest.raises(ValueError, match="shape"):          # same layer names, different input width (3 vs 9 channels)
        C.restore(pkg.dequantization_net.model(device=torch.device("cpu")), prefix, verify=False)
    with pytest.raises(KeyError, match="absent"):
        C.restore(pkg.hallucination_net.model(device=torch.device("cpu")), prefix, verify=False)


def test_checkpoint_initialization_mirrors_the_reference_helper(tmp_path):
    """tf_utils.checkpoint_initialization (tf_utils.py:149-169) + the save loop of joint_training.py:257-263"""
    tu = pkg.tf_utils
    torch.manual_seed(5)
    m = pkg.refinement_net.model(device=torch.device("cpu"))
    d = str(tmp_path / "checkpoints" / "ref")
    ckpt, mgr = tu.checkpoint_initialization("ref", d, m, None)
    assert os.path.isdir(d) and mgr.latest_checkpoint is None and int(ckpt.epoch) == 0
    for _ in range(3):
        ckpt.epoch.assign_add(1)
    path = mgr.save()
    assert path.endswith("ckpt-1") and mgr.latest_checkpoint == path
    ckpt.epoch.assign_add(1)
    assert mgr.save().endswith("ckpt-2")
    torch.manual_seed(6)
    m2 = pkg.refinement_net.model(device=torch.device("cpu"))
    ckpt2, mgr2 = tu.checkpoint_initialization("ref", d, m2, None)          # restores ckpt-2
    assert int(ckpt2.epoch) == 4 and mgr2.save().endswith("ckpt-3")
    for (n, a, _), (_, b, _) in zip(m.named_weights(), m2.named_weights()):
        assert torch.equal(a, b), n

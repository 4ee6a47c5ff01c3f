"""TensorFlow-2 object-graph checkpoints without TensorFlow.

The reference saves and restores every network with
    tf.train.Checkpoint(epoch=tf.Variable(0), lin=model, optimizer=Adam) + CheckpointManager
(tf_utils.py:149-169, test_real_refinement.py:53-68): a directory holding the text state file `checkpoint`,
`ckpt-N.index` and `ckpt-N.data-00000-of-00001`.  This module reads and writes that on-disk format so that
pretrained reference weights drive the HIP path, and so that a training run here can be resumed there:

  * `.index` is a LevelDB-style sorted string table (prefix-compressed blocks + restart array, 1-byte
    compression tag + masked CRC-32C per block, 48-byte footer with magic 0xdb4775248b80fb57) mapping
    key -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}; key "" -> BundleHeaderProto.
  * `.data-*` holds the raw little-endian tensor bytes at those offsets.
  * key `_CHECKPOINTABLE_OBJECT_GRAPH` is a serialized TrackableObjectGraph: nodes with named children and
    attributes {name: "VARIABLE_VALUE", checkpoint_key}.  Variables are located by WALKING that graph along the
    attribute path of the model mirror (`lin -> crf_feature_net -> res1 -> conv1 -> kernel`), so both the
    attribute-named and the `layer_with_weights-N` spellings of a path resolve.

Parity status: TensorFlow is not installable here, so the format is restated from its
published specification (tensorflow/core/util/tensor_bundle, core/lib/io/format.cc, core/protobuf/
trackable_object_graph.proto, tensor_bundle.proto) and pinned only by the CRC-32C / LevelDB known answers and by
round trips through this module's own writer -- "parity unpinned" for this row until a TF-written fixture exists.
Snappy-compressed index blocks (never written by TF's BundleWriter) are rejected with a clear error.
"""
import ctypes
import os
import re
import struct

import numpy as np

try:
    from . import _lib
except ImportError:
    import _lib

TABLE_MAGIC = 0xdb4775248b80fb57
HEADER_KEY = b""
OBJECT_GRAPH_KEY = b"_CHECKPOINTABLE_OBJECT_GRAPH"
VARIABLE_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
# tensorflow/core/framework/types.proto
DT_FLOAT, DT_INT32, DT_STRING, DT_INT64 = 1, 3, 7, 9
_NP_OF = {DT_FLOAT: np.dtype("<f4"), DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8")}
_DT_OF = {np.dtype("float32"): DT_FLOAT, np.dtype("int32"): DT_INT32, np.dtype("int64"): DT_INT64}


# ---------------------------------------------------------------------------
# CRC-32C (libshdr) and LevelDB's mask
# ---------------------------------------------------------------------------
def crc32c(data, crc=0):
    data = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    buf = (ctypes.c_char * len(data)).from_buffer_copy(data) if len(data) else None
    return int(_lib.load().shdr_crc32c(buf, len(data), crc))


def crc32c_array(arr, crc=0):
    arr = np.ascontiguousarray(arr)
    return int(_lib.load().shdr_crc32c(ctypes.c_void_p(arr.ctypes.data), arr.nbytes, crc))


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


# ---------------------------------------------------------------------------
# protobuf wire format (the three message types used here need only varint / length-delimited / fixed32)
# ---------------------------------------------------------------------------
def _put_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError("malformed varint")


def _fields(buf):
    """yield (field_number, wire_type, value) of one serialized message"""
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _get_varint(buf, pos)
        fno, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _get_varint(buf, pos)
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            val = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fno, wt, val


def _f_varint(fno, v):
    return _put_varint(fno << 3) + _put_varint(v)


def _f_bytes(fno, b):
    return _put_varint((fno << 3) | 2) + _put_varint(len(b)) + b


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


class BundleEntry:
    """tensor_bundle.proto BundleEntryProto: 1 dtype, 2 shape{2 dim{1 size}}, 3 shard_id, 4 offset, 5 size, 6 crc32c"""

    def __init__(self, dtype, shape, offset, size, crc, shard_id=0):
        self.dtype, self.shape, self.offset, self.size, self.crc32c, self.shard_id = dtype, tuple(shape), offset, size, crc, shard_id

    def serialize(self):
        dims = b"".join(_f_bytes(2, _f_varint(1, d)) for d in self.shape)
        out = _f_varint(1, self.dtype) + _f_bytes(2, dims)
        if self.shard_id:
            out += _f_varint(3, self.shard_id)
        if self.offset:
            out += _f_varint(4, self.offset)
        out += _f_varint(5, self.size)
        return out + _put_varint((6 << 3) | 5) + struct.pack("<I", self.crc32c)

    @classmethod
    def parse(cls, buf):
        dtype = shard = offset = size = crc = 0
        shape = []
        for fno, _, val in _fields(buf):
            if fno == 1:
                dtype = val
            elif fno == 2:
                for f2, _, dim in _fields(val):
                    if f2 == 2:
                        sz = 0
                        for f3, _, v in _fields(dim):
                            if f3 == 1:
                                sz = _signed64(v)
                        shape.append(sz)
                    elif f2 == 3 and dim:
                        raise ValueError("tensor of unknown rank in checkpoint")
            elif fno == 3:
                shard = val
            elif fno == 4:
                offset = val
            elif fno == 5:
                size = val
            elif fno == 6:
                crc = val
            elif fno == 7:
                raise ValueError("sliced (partitioned) variables are not supported")
        return cls(dtype, shape, offset, size, crc, shard)


# ---------------------------------------------------------------------------
# sorted string table (tensorflow/core/lib/io/{table_builder,format,block}.cc = LevelDB's table format)
# ---------------------------------------------------------------------------
def _parse_block(buf):
    """entries of one block: [shared varint32][non_shared varint32][value_len varint32][key delta][value]...,
    then uint32 restart offsets and their count"""
    num_restarts = struct.unpack_from("<I", buf, len(buf) - 4)[0]
    end = len(buf) - 4 - 4 * num_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _get_varint(buf, pos)
        non_shared, pos = _get_varint(buf, pos)
        vlen, pos = _get_varint(buf, pos)
        key = key[:shared] + bytes(buf[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(buf[pos:pos + vlen])))
        pos += vlen
    return out


def _read_block(data, offset, size, verify):
    contents = data[offset:offset + size]
    ctype = data[offset + size]
    if verify:
        want = struct.unpack_from("<I", data, offset + size + 1)[0]
        if mask_crc(crc32c(data[offset:offset + size + 1])) != want:
            raise ValueError("index block at offset %d fails its CRC-32C" % offset)
    if ctype != 0:
        raise NotImplementedError("compressed index block (type %d); TF's BundleWriter writes uncompressed tables" % ctype)
    return contents


def read_table(path, verify=True):
    """{key: value} of an .index file, in key order"""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint index (bad table magic)" % path)
    footer = data[-48:]
    _, p = _get_varint(footer, 0)           # metaindex handle (unused)
    _, p = _get_varint(footer, p)
    ioff, p = _get_varint(footer, p)
    isize, p = _get_varint(footer, p)
    out = {}
    for _, handle in _parse_block(_read_block(data, ioff, isize, verify)):
        boff, q = _get_varint(handle, 0)
        bsize, q = _get_varint(handle, q)
        for k, v in _parse_block(_read_block(data, boff, bsize, verify)):
            out[k] = v
    return out


class _BlockBuilder:
    def __init__(self, restart_interval):
        self.interval, self.buf, self.restarts, self.count, self.last = restart_interval, bytearray(), [0], 0, b""

    def add(self, key, value):
        shared = 0
        if self.count < self.interval:
            lim = min(len(key), len(self.last))
            while shared < lim and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self):
        return not self.buf


def _shortest_separator(a, b):
    """LevelDB BytewiseComparator::FindShortestSeparator: a key k with a <= k < b"""
    n = min(len(a), len(b))
    i = 0
    while i < n and a[i] == b[i]:
        i += 1
    if i < n and a[i] < 0xff and a[i] + 1 < b[i]:
        return a[:i] + bytes([a[i] + 1])
    return a


def write_table(path, items, block_size=4096):
    """write sorted (key, value) pairs as an uncompressed table (restart interval 16; 1 in the index block)"""
    out = bytearray()
    index = _BlockBuilder(1)
    pending = None              # (last key of the finished block, offset, size)

    def emit(block_bytes):
        off = len(out)
        out.extend(block_bytes)
        out.append(0)                                                # kNoCompression
        out.extend(struct.pack("<I", mask_crc(crc32c(block_bytes + b"\x00"))))
        return off, len(block_bytes)

    blk = _BlockBuilder(16)
    prev = None
    for key, value in items:
        if prev is not None and key <= prev:
            raise ValueError("write_table: keys must be strictly increasing")
        if pending is not None:
            index.add(_shortest_separator(pending[0], key), _put_varint(pending[1]) + _put_varint(pending[2]))
            pending = None
        blk.add(key, value)
        prev = key
        if blk.size() >= block_size:
            off, size = emit(blk.finish())
            pending = (key, off, size)
            blk = _BlockBuilder(16)
    if not blk.empty():
        off, size = emit(blk.finish())
        pending = (prev, off, size)
    if pending is not None:     # LevelDB would use FindShortSuccessor; the full last key is an equally valid bound
        index.add(pending[0], _put_varint(pending[1]) + _put_varint(pending[2]))
    moff, msize = emit(_BlockBuilder(1).finish())                    # empty metaindex block
    ioff, isize = emit(index.finish())
    footer = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    with open(path, "wb") as f:
        f.write(bytes(out))


# ---------------------------------------------------------------------------
# trackable_object_graph.proto
#   TrackableObjectGraph { repeated TrackableObject nodes = 1 }
#   TrackableObject { repeated ObjectReference children = 1 {node_id = 1, local_name = 2};
#                     repeated SerializedTensor attributes = 2 {name = 1, full_name = 2, checkpoint_key = 3};
#                     repeated SlotVariableReference slot_variables = 3 {original_variable_node_id = 1, slot_name = 2,
#                                                                        slot_variable_node_id = 3} }
# ---------------------------------------------------------------------------
class GraphNode:
    def __init__(self):
        self.children, self.attributes, self.slots = {}, {}, []     # name -> node id; name -> checkpoint key


def parse_object_graph(buf):
    nodes = []
    for fno, _, node_buf in _fields(buf):
        if fno != 1:
            continue
        node = GraphNode()
        for f2, _, val in _fields(node_buf):
            if f2 == 1:
                nid, name = 0, ""
                for f3, _, v in _fields(val):
                    if f3 == 1:
                        nid = v
                    elif f3 == 2:
                        name = v.decode()
                node.children[name] = nid
            elif f2 == 2:
                name = key = ""
                for f3, _, v in _fields(val):
                    if f3 == 1:
                        name = v.decode()
                    elif f3 == 3:
                        key = v.decode()
                node.attributes[name] = key
            elif f2 == 3:
                orig = slot = 0
                sname = ""
                for f3, _, v in _fields(val):
                    if f3 == 1:
                        orig = v
                    elif f3 == 2:
                        sname = v.decode()
                    elif f3 == 3:
                        slot = v
                node.slots.append((orig, sname, slot))
        nodes.append(node)
    return nodes


def serialize_object_graph(nodes):
    out = b""
    for node in nodes:
        body = b""
        for name, nid in node.children.items():
            body += _f_bytes(1, (_f_varint(1, nid) if nid else b"") + _f_bytes(2, name.encode()))
        for name, key in node.attributes.items():
            body += _f_bytes(2, _f_bytes(1, name.encode()) + _f_bytes(3, key.encode()))
        for orig, sname, slot in node.slots:
            body += _f_bytes(3, _f_varint(1, orig) + _f_bytes(2, sname.encode()) + _f_varint(3, slot))
        out += _f_bytes(1, body)
    return out


# ---------------------------------------------------------------------------
# reader
# ---------------------------------------------------------------------------
def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: parse the `checkpoint` state file (text CheckpointState proto)"""
    state = os.path.join(directory, "checkpoint")
    if not os.path.exists(state):
        return None
    with open(state) as f:
        m = re.search(r'^model_checkpoint_path:\s*"(.*)"\s*$', f.read(), re.M)
    if not m:
        return None
    path = m.group(1)
    return path if os.path.isabs(path) else os.path.join(directory, path)


class BundleReader:
    def __init__(self, prefix, verify=True):
        self.prefix, self.verify = prefix, verify
        table = read_table(prefix + ".index", verify)
        if HEADER_KEY not in table:
            raise ValueError("checkpoint index has no bundle header")
        self.num_shards = 1
        for fno, _, val in _fields(table[HEADER_KEY]):    # BundleHeaderProto: 1 num_shards, 2 endianness, 3 version
            if fno == 1:
                self.num_shards = val
            elif fno == 2 and val != 0:
                raise NotImplementedError("big-endian checkpoint")
        self.entries = {k.decode(): BundleEntry.parse(v) for k, v in table.items() if k != HEADER_KEY}
        self._files = {}

    def keys(self):
        return sorted(self.entries)

    def _raw(self, e):
        f = self._files.get(e.shard_id)
        if f is None:
            f = self._files[e.shard_id] = open("%s.data-%05d-of-%05d" % (self.prefix, e.shard_id, self.num_shards), "rb")
        f.seek(e.offset)
        raw = f.read(e.size)
        if len(raw) != e.size:
            raise ValueError("checkpoint data file is truncated")
        return raw

    def read(self, key):
        e = self.entries[key]
        raw = self._raw(e)
        if e.dtype == DT_STRING:
            # [varint64 length]*n, 4-byte masked CRC of the lengths, then the bytes (tensor_bundle.cc WriteStringTensor)
            n = int(np.prod(e.shape)) if e.shape else 1
            pos, lens = 0, []
            for _ in range(n):
                ln, pos = _get_varint(raw, pos)
                lens.append(ln)
            pos += 4
            vals = []
            for ln in lens:
                vals.append(raw[pos:pos + ln])
                pos += ln
            return vals[0] if not e.shape else vals
        if e.dtype not in _NP_OF:
            raise NotImplementedError("%s: dtype enum %d" % (key, e.dtype))
        if self.verify and mask_crc(crc32c(raw)) != e.crc32c:
            raise ValueError("%s fails its CRC-32C" % key)
        return np.frombuffer(raw, dtype=_NP_OF[e.dtype]).reshape(e.shape).copy()

    def object_graph(self):
        return parse_object_graph(self.read(OBJECT_GRAPH_KEY.decode()))

    def close(self):
        for f in self._files.values():
            f.close()
        self._files = {}


def _layer_aliases(layer):
    """Keras also links the weighted sub-layers of a Model / Layer as `layer_with_weights-N` (creation order)"""
    out, k = {}, 0
    for name, child in layer._children:
        if child.named_weights():
            out[name] = "layer_with_weights-%d" % k
            k += 1
    return out


def _walk(nodes, nid, layer, prefix, found, missing):
    node = nodes[nid]
    for vname, tensor, _ in layer._vars:
        child = node.children.get(vname)
        key = nodes[child].attributes.get("VARIABLE_VALUE") if child is not None else None
        (found.__setitem__(prefix + vname, (key, child)) if key else missing.append(prefix + vname))
    alias = _layer_aliases(layer)
    for cname, child_layer in layer._children:
        if not child_layer.named_weights():
            continue
        cid = node.children.get(cname, node.children.get(alias.get(cname, "")))
        if cid is None:
            missing.extend(prefix + cname + "." + n for n, _, _ in child_layer.named_weights())
        else:
            _walk(nodes, cid, child_layer, prefix + cname + ".", found, missing)


def restore(model, path, root="lin", optimizer=None, verify=True, strict=True):
    """Load the variables of `model` (a singlehdr-tf2_amd network) from a reference checkpoint.

    `path`: a checkpoint prefix (".../ckpt-5") or a CheckpointManager directory.  `root`: the keyword the model was
    saved under (the reference always uses `lin`, tf_utils.py:158).  `optimizer`: a pipeline.KerasAdam whose FlatParams
    hold exactly this model's variables -- its m / v slots and step count are restored when the checkpoint has them.
    Returns {"epoch": int | None, "restored": [names], "missing": [names]}."""
    import torch
    prefix = latest_checkpoint(path) if os.path.isdir(path) else path
    if prefix is None:
        raise FileNotFoundError("no checkpoint state file in %s" % path)
    rd = BundleReader(prefix, verify)
    try:
        nodes = rd.object_graph()
        if root not in nodes[0].children:
            raise KeyError("checkpoint has no object %r (children: %s)" % (root, sorted(nodes[0].children)))
        found, missing = {}, []
        _walk(nodes, nodes[0].children[root], model, "", found, missing)
        if strict and missing:
            raise KeyError("variables absent from the checkpoint: %s" % missing[:8])
        own = {n: t for n, t, _ in model.named_weights()}
        with torch.no_grad():
            for name, (key, _) in found.items():
                arr = rd.read(key)
                t = own[name]
                if tuple(arr.shape) != tuple(t.shape):
                    raise ValueError("%s: checkpoint shape %s != model %s" % (name, arr.shape, tuple(t.shape)))
                t.copy_(torch.from_numpy(arr.astype(np.float32)))
        epoch = None
        ep = nodes[0].children.get("epoch")
        if ep is not None and "VARIABLE_VALUE" in nodes[ep].attributes:
            epoch = int(rd.read(nodes[ep].attributes["VARIABLE_VALUE"]).reshape(-1)[0])
        if optimizer is not None and "optimizer" in nodes[0].children:
            onode = nodes[nodes[0].children["optimizer"]]
            by_var = {}
            for orig, sname, slot in onode.slots:
                by_var.setdefault(orig, {})[sname] = nodes[slot].attributes.get("VARIABLE_VALUE")
            fp = optimizer.p
            name_of = {id(t): n for n, t in own.items()}
            with torch.no_grad():
                for v, off in zip(fp.variables, fp.offsets):
                    name = name_of.get(id(v))
                    slots = by_var.get(found[name][1], {}) if name in found else {}
                    for sname, buf in (("m", fp.m), ("v", fp.v)):
                        if slots.get(sname):
                            buf[off:off + v.numel()].copy_(torch.from_numpy(rd.read(slots[sname]).astype(np.float32)).reshape(-1))
            it = onode.children.get("iter")
            if it is not None and "VARIABLE_VALUE" in nodes[it].attributes:
                optimizer.t = int(rd.read(nodes[it].attributes["VARIABLE_VALUE"]).reshape(-1)[0])
        return dict(epoch=epoch, restored=sorted(found), missing=missing, prefix=prefix)
    finally:
        rd.close()


# ---------------------------------------------------------------------------
# writer
# ---------------------------------------------------------------------------
def _string_tensor_bytes(value):
    """scalar DT_STRING payload and its bundle CRC (tensor_bundle.cc WriteStringTensor)"""
    lens = _put_varint(len(value))
    crc = crc32c(struct.pack("<I", len(value)))
    cks = struct.pack("<I", mask_crc(crc))
    crc = crc32c(value, crc32c(cks, crc))
    return lens + cks + value, crc


def save(directory, model, root="lin", epoch=0, optimizer=None, save_counter=1, max_to_keep=5):
    """Write `ckpt-<save_counter>` in the reference's layout (tf_utils.py:149-169): objects `epoch`, `<root>` (the model),
    `optimizer` (iter + Adam m / v slots when given) and `save_counter`; updates the `checkpoint` state file."""
    os.makedirs(directory, exist_ok=True)
    nodes = [GraphNode()]
    tensors = {}                                   # checkpoint key -> ndarray | bytes

    def new_node():
        nodes.append(GraphNode())
        return len(nodes) - 1

    def add_variable(parent, name, path, arr):
        nid = new_node()
        nodes[parent].children[name] = nid
        key = path + VARIABLE_SUFFIX
        nodes[nid].attributes["VARIABLE_VALUE"] = key
        tensors[key] = arr
        return nid

    add_variable(0, "epoch", "epoch", np.asarray(epoch, dtype=np.int32))
    var_node = {}

    def add_layer(parent, name, path, layer):
        nid = new_node()
        nodes[parent].children[name] = nid
        for vname, t, _ in layer._vars:
            var_node[id(t)] = (add_variable(nid, vname, path + "/" + vname, t.detach().cpu().numpy().astype(np.float32)), path + "/" + vname)
        for cname, child in layer._children:
            if child.named_weights():
                add_layer(nid, cname, path + "/" + cname, child)
        for cname, alias in _layer_aliases(layer).items():          # Keras' second name for the same node
            nodes[nid].children[alias] = nodes[nid].children[cname]
        return nid

    add_layer(0, root, root, model)
    if optimizer is not None:
        oid = new_node()
        nodes[0].children["optimizer"] = oid
        add_variable(oid, "iter", "optimizer/iter", np.asarray(optimizer.t, dtype=np.int64))
        for hname, val in (("learning_rate", optimizer.lr), ("beta_1", optimizer.b1), ("beta_2", optimizer.b2), ("decay", 0.0)):
            add_variable(oid, hname, "optimizer/" + hname, np.asarray(val, dtype=np.float32))
        fp = optimizer.p
        for v, off in zip(fp.variables, fp.offsets):
            if id(v) not in var_node:
                continue
            vid, vpath = var_node[id(v)]
            for sname, buf in (("m", fp.m), ("v", fp.v)):
                sid = new_node()
                key = "%s/.OPTIMIZER_SLOT/optimizer/%s%s" % (vpath, sname, VARIABLE_SUFFIX)
                nodes[sid].attributes["VARIABLE_VALUE"] = key
                tensors[key] = buf[off:off + v.numel()].detach().cpu().numpy().reshape(tuple(v.shape)).astype(np.float32)
                nodes[oid].slots.append((vid, sname, sid))
    add_variable(0, "save_counter", "save_counter", np.asarray(save_counter, dtype=np.int64))
    tensors[OBJECT_GRAPH_KEY.decode()] = serialize_object_graph(nodes)

    prefix = os.path.join(directory, "ckpt-%d" % save_counter)
    items = [(HEADER_KEY, _f_varint(1, 1) + _f_bytes(3, _f_varint(1, 1)))]     # num_shards = 1, version.producer = 1
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for key in sorted(tensors, key=lambda k: k.encode()):
            val = tensors[key]
            if isinstance(val, bytes):
                payload, crc = _string_tensor_bytes(val)
                entry = BundleEntry(DT_STRING, (), offset, len(payload), mask_crc(crc))
            else:
                arr = np.ascontiguousarray(val)
                payload = arr.tobytes()
                entry = BundleEntry(_DT_OF[arr.dtype], arr.shape, offset, len(payload), mask_crc(crc32c_array(arr)))
            f.write(payload)
            offset += len(payload)
            items.append((key.encode(), entry.serialize()))
    write_table(prefix + ".index", items)
    # CheckpointManager state file
    state = os.path.join(directory, "checkpoint")
    kept = []
    if os.path.exists(state):
        with open(state) as f:
            kept = re.findall(r'^all_model_checkpoint_paths:\s*"(.*)"\s*$', f.read(), re.M)
    name = os.path.basename(prefix)
    kept = [k for k in kept if k != name] + [name]
    for old in kept[:-max_to_keep] if max_to_keep else []:
        for suffix in (".index", ".data-00000-of-00001"):
            try:
                os.remove(os.path.join(directory, old + suffix))
            except OSError:
                pass
    kept = kept[-max_to_keep:] if max_to_keep else kept
    with open(state, "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % name)
        for k in kept:
            f.write('all_model_checkpoint_paths: "%s"\n' % k)
    return prefix

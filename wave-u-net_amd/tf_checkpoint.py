"""Reader / writer for TensorFlow "V2" checkpoints (tensor bundles), without TensorFlow.

The reference saves and restores with `tf.train.Saver(..., write_version=SaverDef.V2)` (`Training.py:92-98,113`,
`Evaluate.py:55-57`); the published weights (`README.md:110-111`) are such checkpoints.  A V2 checkpoint `<prefix>` is
two files:

  <prefix>.index                 an SSTable (the LevelDB table format: prefix-compressed key/value blocks, an index
                                 block, a 48-byte footer) mapping tensor name -> BundleEntryProto {dtype, shape, shard,
                                 offset, size, masked crc32c}; key "" holds the BundleHeaderProto
  <prefix>.data-00000-of-00001   the raw little-endian tensor bytes

This module restates that published format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table*,
tensorflow/core/protobuf/tensor_bundle.proto, TF 1.8): `read(prefix)` returns {name: ndarray}, `write(prefix, tensors)`
produces a checkpoint `tf.train.Saver.restore` / `tf.train.load_checkpoint` accept.  TensorFlow is not installed in the
build image, so the pair is checked against each other and against the format's published constants (table magic,
masked CRC-32C, known CRC vectors) -- not against a file written by TensorFlow; `tests/test_tf_checkpoint.py` says so.

The mapping between these tensors and a separator's arenas is in `checkpoint.py`.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
_CRC_POLY = 0x82F63B78          # CRC-32C (Castagnoli), reflected
_MASK_DELTA = 0xA282EAD8

# tensorflow/core/framework/types.proto
_DT = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"),
       6: np.dtype("i1"), 9: np.dtype("<i8"), 10: np.dtype("bool"), 17: np.dtype("<u2"), 19: np.dtype("<f2"),
       22: np.dtype("<u4"), 23: np.dtype("<u8")}
_DT_OF = {v: k for k, v in _DT.items()}


# ------------------------------------------------------------------------------------------------ CRC-32C
def _make_table():
    t = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ _CRC_POLY if c & 1 else c >> 1
        t[i] = c
    return t


_TABLE = _make_table()
_TABLE_PY = [int(x) for x in _TABLE]


def _crc_small(data, crc=0):
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = _TABLE_PY[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _gf2_times(mat, vec):
    s, i = 0, 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1
        i += 1
    return s


def _gf2_square(mat):
    return [_gf2_times(mat, mat[n]) for n in range(32)]


def _zeros_matrix(nbytes):
    """32x32 GF(2) operator (as 32 column words) that advances a CRC register over `nbytes` zero bytes
    (the crc32_combine construction)."""
    odd = [_CRC_POLY] + [1 << (n - 1) for n in range(1, 32)]      # one zero bit
    even = _gf2_square(odd)                                        # two
    odd = _gf2_square(even)                                        # four
    cols = [1 << n for n in range(32)]                             # identity
    n = nbytes
    while True:
        even = _gf2_square(odd)                                    # first pass: 8 bits = one byte
        if n & 1:
            cols = [_gf2_times(even, c) for c in cols]
        n >>= 1
        if not n:
            break
        odd = _gf2_square(even)
        if n & 1:
            cols = [_gf2_times(odd, c) for c in cols]
        n >>= 1
        if not n:
            break
    return cols


def crc32c(data):
    """CRC-32C of a bytes-like object.  Large buffers are cut into equal chunks whose CRCs advance together as one
    numpy vector, then folded with the zero-operator of the chunk length (pure Python would take a minute for a
    100 MB checkpoint)."""
    buf = np.frombuffer(memoryview(data).cast("B"), dtype=np.uint8)
    n = buf.size
    if n < (1 << 16):
        return _crc_small(buf.tobytes())
    lanes = 4096
    L = n // lanes
    body = np.ascontiguousarray(buf[:lanes * L].reshape(lanes, L).T)       # [L, lanes]: step j is one row
    crc = np.full(lanes, 0xFFFFFFFF, dtype=np.uint32)
    for j in range(L):
        crc = _TABLE[(crc ^ body[j]) & 0xFF] ^ (crc >> 8)
    crc ^= np.uint32(0xFFFFFFFF)
    mat = _zeros_matrix(L)
    total = int(crc[0])
    for c in crc[1:]:
        total = _gf2_times(mat, total) ^ int(c)
    tail = buf[lanes * L:]
    if tail.size:
        total = _crc_small(tail.tobytes(), total)
    return total


def mask_crc(crc):
    """tensorflow/core/lib/hash/crc32c.h: rotate right by 15 and add a constant (CRCs of data that embeds CRCs)."""
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked):
    rot = (masked - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ varints / protobuf
def _get_varint(b, pos):
    x, shift = 0, 0
    while True:
        byte = b[pos]
        pos += 1
        x |= (byte & 0x7F) << shift
        if not byte & 0x80:
            return x, pos
        shift += 7


def _put_varint(x):
    out = bytearray()
    while True:
        if x < 0x80:
            out.append(x)
            return bytes(out)
        out.append((x & 0x7F) | 0x80)
        x >>= 7


def _pb_fields(b):
    """Yield (field number, wire type, value) of one protobuf message."""
    pos = 0
    while pos < len(b):
        key, pos = _get_varint(b, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(b, pos)
        elif wt == 1:
            v = b[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(b, pos)
            v = b[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = b[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _signed64(x):
    return x - (1 << 64) if x >= (1 << 63) else x


def _parse_entry(b):
    """BundleEntryProto: 1 dtype, 2 shape {2: dim {1: size}}, 3 shard_id, 4 offset, 5 size, 6 crc32c (fixed32), 7 slices."""
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "slices": 0}
    for f, _, v in _pb_fields(b):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            for f2, _, v2 in _pb_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _pb_fields(v2):
                        if f3 == 1:
                            size = _signed64(v3)
                    e["shape"].append(size)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
        elif f == 7:
            e["slices"] += 1
    return e


def _pb_varint_field(f, x):
    return _put_varint((f << 3) | 0) + _put_varint(x & ((1 << 64) - 1))


def _pb_bytes_field(f, b):
    return _put_varint((f << 3) | 2) + _put_varint(len(b)) + b


def _build_entry(dtype, shape, offset, size, crc):
    shp = b"".join(_pb_bytes_field(2, _pb_varint_field(1, int(d))) for d in shape)
    out = _pb_varint_field(1, dtype) + _pb_bytes_field(2, shp)
    # proto3: zero-valued scalars are omitted (shard_id 0, offset 0)
    if offset:
        out += _pb_varint_field(4, offset)
    if size:
        out += _pb_varint_field(5, size)
    out += _put_varint((6 << 3) | 5) + struct.pack("<I", crc)
    return out


# ------------------------------------------------------------------------------------------------ snappy (index blocks)
def _snappy_decompress(b):
    n, pos = _get_varint(b, 0)
    out = bytearray()
    while pos < len(b):
        tag = b[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(b[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += b[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = 4 + ((tag >> 2) & 7)
            off = ((tag >> 5) << 8) | b[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(b[pos:pos + 2], "little")
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(b[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy block")
        for _ in range(ln):                       # may overlap its own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("corrupt snappy block (length)")
    return bytes(out)


# ------------------------------------------------------------------------------------------------ SSTable
def _read_block(data, offset, size, verify):
    body = data[offset:offset + size]
    ctype = data[offset + size]
    if verify:
        stored = struct.unpack("<I", data[offset + size + 1:offset + size + 5])[0]
        if unmask_crc(stored) != _crc_small(data[offset:offset + size + 1]):
            raise ValueError("index block checksum mismatch at offset %d" % offset)
    if ctype == 1:
        body = _snappy_decompress(body)
    elif ctype != 0:
        raise ValueError("unknown block compression type %d" % ctype)
    return body


def _block_entries(block):
    nrestart = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * nrestart
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _table_entries(data, verify=True):
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != TABLE_MAGIC:
        raise ValueError("not a TensorFlow checkpoint index (bad table magic)")
    footer = data[-48:]
    _, pos = _get_varint(footer, 0)               # metaindex handle (unused)
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        for kv in _block_entries(_read_block(data, boff, bsize, verify)):
            yield kv


class _BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b""
        self.interval = restart_interval

    def add(self, key, value):
        shared = 0
        if self.count < self.interval:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self):
        out = bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts)
        return out + struct.pack("<I", len(self.restarts))

    def __len__(self):
        return len(self.buf) + 4 * len(self.restarts) + 4


def _build_table(items, block_size=4096):
    """items: sorted [(key bytes, value bytes)] -> the bytes of an uncompressed SSTable."""
    out = bytearray()
    index = _BlockBuilder(restart_interval=1)

    def emit(block_bytes):
        off = len(out)
        out.extend(block_bytes)
        out.append(0)                                             # kNoCompression
        out.extend(struct.pack("<I", mask_crc(_crc_small(block_bytes + b"\x00"))))
        return _put_varint(off) + _put_varint(len(block_bytes))

    cur = _BlockBuilder()
    last_key = None
    for key, value in items:
        if last_key is not None and key <= last_key:
            raise ValueError("table keys must be strictly increasing")
        cur.add(key, value)
        last_key = key
        if len(cur) >= block_size:
            index.add(last_key, emit(cur.finish()))
            cur = _BlockBuilder()
    if cur.count or not items:
        index.add(last_key if last_key is not None else b"", emit(cur.finish()))
    meta_handle = emit(_BlockBuilder().finish())                  # empty metaindex block
    index_handle = emit(index.finish())
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    return bytes(out)


# ------------------------------------------------------------------------------------------------ public API
def is_checkpoint(prefix):
    return os.path.isfile(str(prefix) + ".index")


def list_variables(prefix):
    """[(name, dtype, shape)] like tf.train.list_variables."""
    with open(str(prefix) + ".index", "rb") as f:
        data = f.read()
    out = []
    for key, value in _table_entries(data):
        if key == b"":
            continue
        e = _parse_entry(value)
        out.append((key.decode("utf-8"), _DT.get(e["dtype"]), tuple(e["shape"])))
    return out


def read(prefix, verify=True):
    """{tensor name: ndarray} of a V2 checkpoint.  `verify` checks the CRC-32C of the index blocks and of every
    tensor (as the TensorFlow reader does)."""
    prefix = str(prefix)
    with open(prefix + ".index", "rb") as f:
        data = f.read()
    shards = {}
    num_shards = 1
    out = {}
    for key, value in _table_entries(data, verify):
        if key == b"":
            for f_, _, v in _pb_fields(value):                    # BundleHeaderProto: 1 num_shards, 2 endianness
                if f_ == 1:
                    num_shards = v
                elif f_ == 2 and v != 0:
                    raise ValueError("big-endian checkpoints are not supported")
            continue
        e = _parse_entry(value)
        name = key.decode("utf-8")
        if e["slices"]:
            raise ValueError("%s: partitioned (sliced) variables are not supported" % name)
        if e["dtype"] not in _DT:
            raise ValueError("%s: unsupported dtype enum %d" % (name, e["dtype"]))
        sid = e["shard_id"]
        if sid not in shards:
            with open("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), "rb") as f:
                shards[sid] = f.read()
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        dt = _DT[e["dtype"]]
        count = int(np.prod(e["shape"])) if e["shape"] else 1
        if len(raw) != e["size"] or count * dt.itemsize != e["size"]:
            raise ValueError("%s: size %d does not match shape %s" % (name, e["size"], e["shape"]))
        if verify and e["crc32c"] is not None and unmask_crc(e["crc32c"]) != crc32c(raw):
            raise ValueError("%s: tensor checksum mismatch" % name)
        out[name] = np.frombuffer(raw, dtype=dt).reshape(e["shape"]).copy()
    return out


def write(prefix, tensors):
    """Write {name: array} as a one-shard V2 checkpoint (`<prefix>.index`, `<prefix>.data-00000-of-00001`)."""
    prefix = str(prefix)
    items = []
    blob = bytearray()
    for name in sorted(tensors, key=lambda s: s.encode("utf-8")):
        arr = np.asarray(tensors[name])
        dt = arr.dtype.newbyteorder("<") if arr.dtype.byteorder == ">" else arr.dtype
        if np.dtype(dt) not in _DT_OF:
            raise ValueError("%s: dtype %s has no TensorFlow enum here" % (name, arr.dtype))
        raw = np.ascontiguousarray(arr, dtype=dt).tobytes()
        entry = _build_entry(_DT_OF[np.dtype(dt)], arr.shape, len(blob), len(raw), mask_crc(crc32c(raw)))
        items.append((name.encode("utf-8"), entry))
        blob += raw
    # BundleHeaderProto {num_shards = 1, endianness = LITTLE (0, omitted), version {producer = 1}}
    header = _pb_varint_field(1, 1) + _pb_bytes_field(3, _pb_varint_field(1, 1))
    table = _build_table([(b"", header)] + items)
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(blob))
    with open(prefix + ".index", "wb") as f:
        f.write(table)
    return prefix

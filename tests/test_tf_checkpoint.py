"""TensorFlow V2 checkpoint reader / writer (wave_u_net_amd/tf_checkpoint.py) -- CPU only.

TensorFlow is not installed in the build image, so nothing here reads a file that TensorFlow wrote: the writer and the
reader are checked against each other, against the published constants of the format (CRC-32C check value, the
masked-CRC definition, the table magic) and against a hand-assembled index block that uses what the writer never
emits (snappy-compressed blocks, a shard_id field, multi-entry prefix compression)."""
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wave_u_net_amd import tf_checkpoint as tfc


def test_crc32c_known_answers():
    # the CRC-32C check value (RFC 3720 appendix B.4 lists the same polynomial's vectors)
    assert tfc.crc32c(b"123456789") == 0xE3069283
    assert tfc.crc32c(bytes(32)) == 0x8A9136AA
    assert tfc.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tfc.crc32c(bytes(range(32))) == 0x46DD794E
    assert tfc.crc32c(b"") == 0


def test_crc32c_vector_path_equals_byte_loop():
    rng = np.random.default_rng(3)
    for n in (1 << 16, (1 << 16) + 1, 300001, 4096 * 17 + 4095):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert tfc.crc32c(data) == tfc._crc_small(data), n


def test_masked_crc_definition():
    # tensorflow/core/lib/hash/crc32c.h: Mask(crc) = ((crc >> 15) | (crc << 17)) + 0xa282ead8
    crc = tfc.crc32c(b"foo")
    assert tfc.mask_crc(crc) != crc
    assert tfc.unmask_crc(tfc.mask_crc(crc)) == crc
    assert tfc.mask_crc(0) == 0xA282EAD8


def _tensors(rng):
    return {
        "separator/conv1d/kernel": rng.standard_normal((15, 1, 24)).astype(np.float32),
        "separator/conv1d/bias": rng.standard_normal(24).astype(np.float32),
        "separator/conv1d_1/kernel": rng.standard_normal((15, 24, 48)).astype(np.float32),
        "separator/interp_0": rng.standard_normal((24,)).astype(np.float32),
        "global_step": np.asarray(123456789012, dtype=np.int64),
        "separator_solver/beta1_power": np.asarray(0.9 ** 7, dtype=np.float32),
        "empty": np.zeros((0, 3), dtype=np.float32),
        "big": rng.standard_normal((300, 400)).astype(np.float32),          # > 64 KiB: the vectorised CRC path
    }


def test_write_read_round_trip(tmp_path):
    t = _tensors(np.random.default_rng(0))
    prefix = tfc.write(tmp_path / "ck" / "model-7", t)
    assert os.path.isfile(prefix + ".index") and os.path.isfile(prefix + ".data-00000-of-00001")
    assert tfc.is_checkpoint(prefix)
    back = tfc.read(prefix)
    assert sorted(back) == sorted(t)
    for k in t:
        assert back[k].dtype == t[k].dtype and back[k].shape == t[k].shape, k
        assert np.array_equal(back[k], t[k]), k
    listed = {n: (d, s) for n, d, s in tfc.list_variables(prefix)}
    assert listed["separator/conv1d_1/kernel"] == (np.dtype("<f4"), (15, 24, 48))
    assert listed["global_step"] == (np.dtype("<i8"), ())
    # layout of the index file: 48-byte footer ending in the table magic; the data file is the tensors back to back
    idx = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", idx[-8:])[0] == 0xDB4775248B80FB57
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(v.nbytes for v in t.values())


def test_many_variables_span_several_index_blocks(tmp_path):
    rng = np.random.default_rng(1)
    t = {"separator/conv1d_%d/kernel" % i: rng.standard_normal((3, 2)).astype(np.float32) for i in range(400)}
    prefix = tfc.write(tmp_path / "m", t)
    assert os.path.getsize(prefix + ".index") > 3 * 4096             # several 4 KiB data blocks + the index block
    back = tfc.read(prefix)
    assert len(back) == 400 and all(np.array_equal(back[k], t[k]) for k in t)


def test_corruption_is_detected(tmp_path):
    t = _tensors(np.random.default_rng(2))
    prefix = tfc.write(tmp_path / "c", t)
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[100] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(ValueError, match="checksum"):
        tfc.read(prefix)
    assert len(tfc.read(prefix, verify=False)) == len(t)              # (what an unchecked read returns is the caller's risk)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[10] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError):
        tfc.read(prefix)
    open(prefix + ".index", "wb").write(b"not a table" * 10)
    with pytest.raises(ValueError, match="magic"):
        tfc.read(prefix)


def _snappy_literal(raw):
    """A valid snappy stream: length preamble + one literal element."""
    tag = bytes([(60 << 2) | 0, len(raw) - 1]) if len(raw) > 60 else bytes([((len(raw) - 1) << 2) | 0])
    return tfc._put_varint(len(raw)) + tag + raw


def test_snappy_decoder_literals_and_copies():
    raw = bytes(range(200))
    assert tfc._snappy_decompress(_snappy_literal(raw)) == raw
    # "abcdefgh" + copy(offset 8, length 8) twice (2-byte-offset form, then the 1-byte-offset form) + overlapping run
    stream = tfc._put_varint(8 + 8 + 8 + 6) + bytes([(7 << 2) | 0]) + b"abcdefgh"
    stream += bytes([(7 << 2) | 2]) + struct.pack("<H", 8)
    stream += bytes([((8 - 4) << 2) | 1, 8])
    stream += bytes([((6 - 4) << 2) | 1, 1])                         # offset 1, length 6: repeats the last byte
    assert tfc._snappy_decompress(stream) == b"abcdefgh" * 3 + b"h" * 6
    with pytest.raises(ValueError):
        tfc._snappy_decompress(stream[:-2] + bytes([((6 - 4) << 2) | 1, 200]))


def test_reads_what_the_writer_never_emits(tmp_path):
    """Hand-assembled index: a snappy-compressed data block and an entry that carries an explicit shard_id field."""
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    b = np.asarray(5, dtype=np.int64)
    blob = a.tobytes() + b.tobytes()
    ea = tfc._build_entry(1, a.shape, 0, a.nbytes, tfc.mask_crc(tfc.crc32c(a.tobytes())))
    ea += tfc._pb_varint_field(3, 0)                                  # explicit shard_id = 0
    eb = tfc._build_entry(9, (), a.nbytes, b.nbytes, tfc.mask_crc(tfc.crc32c(b.tobytes())))
    blk = tfc._BlockBuilder()
    for k, v in ((b"", tfc._pb_varint_field(1, 1)), (b"separator/a", ea), (b"separator/b", eb)):
        blk.add(k, v)
    out = bytearray()

    def emit(body, ctype):
        off = len(out)
        out.extend(body)
        out.append(ctype)
        out.extend(struct.pack("<I", tfc.mask_crc(tfc._crc_small(bytes(body) + bytes([ctype])))))
        return tfc._put_varint(off) + tfc._put_varint(len(body))

    h_data = emit(_snappy_literal(blk.finish()), 1)                   # compression type 1 = snappy
    h_meta = emit(tfc._BlockBuilder().finish(), 0)
    ib = tfc._BlockBuilder(restart_interval=1)
    ib.add(b"separator/c", h_data)
    h_index = emit(ib.finish(), 0)
    footer = h_meta + h_index
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", tfc.TABLE_MAGIC))
    prefix = str(tmp_path / "hand")
    open(prefix + ".index", "wb").write(bytes(out))
    open(prefix + ".data-00000-of-00001", "wb").write(blob)
    back = tfc.read(prefix)
    assert np.array_equal(back["separator/a"], a) and back["separator/b"] == 5


def test_round_trip_of_arbitrary_names_shapes_and_dtypes(tmp_path):
    """Property test (hypothesis): whatever set of tensors is written comes back identical -- names with shared
    prefixes (the index blocks' prefix compression), empty and scalar shapes, every dtype the reference's graph uses."""
    hyp = pytest.importorskip("hypothesis")
    st = pytest.importorskip("hypothesis.strategies")
    name = st.text(alphabet="abcdefgh_/0123456789", min_size=1, max_size=40)
    shape = st.lists(st.integers(min_value=0, max_value=5), min_size=0, max_size=3)
    dtype = st.sampled_from([np.float32, np.int64, np.int32, np.float64])
    counter = [0]

    @hyp.settings(max_examples=40, deadline=None)
    @hyp.given(st.dictionaries(name, st.tuples(shape, dtype, st.integers(0, 2 ** 31 - 1)), min_size=0, max_size=25))
    def check(spec):
        tensors = {}
        for k, (shp, dt, seed) in spec.items():
            rng = np.random.default_rng(seed)
            tensors[k] = (rng.standard_normal(shp) * 100).astype(dt)
        counter[0] += 1
        prefix = tfc.write(tmp_path / ("p%d" % counter[0]) / "ck", tensors)
        back = tfc.read(prefix)
        assert sorted(back) == sorted(tensors)
        for k, v in tensors.items():
            assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k

    check()

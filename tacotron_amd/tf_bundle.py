"""TensorFlow-free reader of a TF checkpoint in the "tensor bundle" (V2) format: `<prefix>.index` + `<prefix>.data-?????-of-?????`
-- the files `download_weights.sh:4-7` unpacks and `saver.restore` reads (train.py:47-58, test.py:40-47).

With it `tf_import.import_tf_variables` needs only the downloaded files, not a TensorFlow install (VERDICT r4 #8b): the
pretrained Nancy checkpoint is the one artefact that could falsify the TF-semantics assumptions of oracle/taco_numpy.py.

Format (restated from the public file-format descriptions of LevelDB tables, doc/table_format.md, and of
tensorflow/core/protobuf/tensor_bundle.proto; no TensorFlow code is used or copied):
  * `.index` is a LevelDB-style sorted table: data blocks, a metaindex block, an index block and a 48-byte footer
    `[metaindex handle][index handle][padding to 40 bytes][magic 0xdb4775248b80fb57, little endian]`; a handle is two varint64
    (offset, size); every block is followed by a 1-byte compression type (0 none, 1 snappy) and a masked CRC-32C of block + type.
    A block is a run of prefix-compressed entries `varint32 shared | varint32 non_shared | varint32 value_len | key delta | value`
    followed by `uint32 restart[n] | uint32 n`.
  * the value of key "" is a BundleHeaderProto {1: num_shards, 2: endianness (0 little), 3: version}; every other key is a
    variable name and its value a BundleEntryProto {1: dtype, 2: shape {2: dim {1: size}}, 3: shard_id, 4: offset, 5: size,
    6: fixed32 masked CRC-32C of the bytes, 7: slices (partitioned variables: not supported here, the reference has none)}.
  * tensor bytes sit at `offset` of shard `shard_id`, little endian, row-major.
STATUS: written from the format descriptions and exercised against an independent writer of the same format in
tests/test_host.py (block layout, prefix compression across restarts, multi-block index, CRCs, snappy-compressed index
blocks); it has not met a file written by TensorFlow itself in this environment (none is reachable)."""
from __future__ import annotations

import os
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}


# ---- CRC-32C (Castagnoli), table driven; the stored values are "masked": rot15(crc) + 0xa282ead8 -----------------------------
def _make_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        t.append(c)
    return t


_T = _make_table()
_T_NP = np.array(_T, dtype=np.uint32)


def crc32c(data: bytes, crc: int = 0) -> int:
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = _T[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c: int) -> int:
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ---- varints / protobuf wire format --------------------------------------------------------------------------------------------
def _varint(buf, pos):
    r = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        r |= (b & 0x7F) << shift
        if b < 0x80:
            return r, pos
        shift += 7
        if shift > 70:
            raise ValueError('malformed varint')


def _fields(buf):
    """Yields (field number, wire type, value) of one serialized message; value = int or bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield f, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


# ---- snappy (raw format) decompression: index blocks may be compressed ---------------------------------------------------------
def snappy_decompress(src: bytes) -> bytes:
    n, pos = _varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += src[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = src[pos] | (src[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError('malformed snappy stream')
        for _ in range(ln):                              # (overlapping copies are the point of the format: byte by byte)
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('snappy: %d bytes decoded, header says %d' % (len(out), n))
    return bytes(out)


# ---- table ------------------------------------------------------------------------------------------------------------------------
def _read_block(f, offset, size, verify=True):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) != size + 5:
        raise ValueError('truncated table: block at %d (+%d)' % (offset, size))
    body, ctype, stored = raw[:size], raw[size], struct.unpack_from('<I', raw, size + 1)[0]
    if verify and mask_crc(crc32c(raw[:size + 1])) != stored:
        raise ValueError('block at %d: CRC mismatch' % offset)
    if ctype == 1:
        body = snappy_decompress(body)
    elif ctype != 0:
        raise ValueError('block at %d: unknown compression type %d' % (offset, ctype))
    return body


def _block_entries(body):
    """(key, value) pairs of one block, prefix compression undone."""
    n_restarts = struct.unpack_from('<I', body, len(body) - 4)[0]
    end = len(body) - 4 - 4 * n_restarts
    pos, key = 0, b''
    while pos < end:
        shared, pos = _varint(body, pos)
        non_shared, pos = _varint(body, pos)
        vlen, pos = _varint(body, pos)
        key = key[:shared] + bytes(body[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(body[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True):
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    with open(path, 'rb') as f:
        f.seek(0, os.SEEK_END)
        size = f.tell()
        if size < 48:
            raise ValueError('%s: too short for a table footer' % path)
        f.seek(size - 48)
        footer = f.read(48)
        if struct.unpack_from('<Q', footer, 40)[0] != MAGIC:
            raise ValueError('%s: not a table file (bad magic)' % path)
        _, p = _varint(footer, 0)
        _, p = _varint(footer, p)                # metaindex handle: unused
        ioff, p = _varint(footer, p)
        isz, p = _varint(footer, p)
        out = []
        for _, handle in _block_entries(_read_block(f, ioff, isz, verify)):
            boff, q = _varint(handle, 0)
            bsz, q = _varint(handle, q)
            out.extend(_block_entries(_read_block(f, boff, bsz, verify)))
        return out


# ---- bundle -----------------------------------------------------------------------------------------------------------------------
def _parse_entry(buf):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'sliced': False}
    for f, wt, v in _fields(buf):
        if f == 1:
            e['dtype'] = v
        elif f == 2:
            for f2, _, v2 in _fields(v):
                if f2 == 2:
                    d = 0
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            d = _signed64(v3)
                    e['shape'].append(d)
        elif f == 3:
            e['shard_id'] = v
        elif f == 4:
            e['offset'] = _signed64(v)
        elif f == 5:
            e['size'] = _signed64(v)
        elif f == 6:
            e['crc32c'] = v
        elif f == 7:
            e['sliced'] = True
    return e


def list_variables(prefix, verify=True):
    """{name: (numpy dtype, shape)} of a checkpoint `<prefix>.index` (no tensor bytes are read)."""
    _, entries = _index(prefix, verify)
    return {k: (_DTYPES.get(e['dtype']), tuple(e['shape'])) for k, e in entries.items()}


def _index(prefix, verify):
    header, entries = None, {}
    for k, v in read_table(prefix + '.index', verify):
        if k == b'':
            header = {f: val for f, _, val in _fields(v)}
        else:
            entries[k.decode('utf-8')] = _parse_entry(v)
    if header is None:
        raise ValueError('%s.index: no bundle header' % prefix)
    if header.get(2, 0) != 0:
        raise ValueError('big-endian bundles are not supported')
    return header, entries


def load_checkpoint(prefix, names=None, verify_tensors=False, verify_index=True):
    """{variable name: numpy array} of the tensor bundle `<prefix>.index` / `<prefix>.data-*` -- what
    `tf.train.load_checkpoint(prefix)` + `get_tensor` give, without TensorFlow.  names: restrict to these variables.
    verify_tensors: check every tensor's CRC-32C (pure Python: ~10 MB/s)."""
    header, entries = _index(prefix, verify_index)
    shards = int(header.get(1, 1))
    files = {}
    out = {}
    try:
        for k, e in entries.items():
            if names is not None and k not in names:
                continue
            if e['sliced']:
                raise ValueError('%s: partitioned (sliced) variables are not supported' % k)
            dt = _DTYPES.get(e['dtype'])
            if dt is None:
                raise ValueError('%s: unsupported dtype enum %d' % (k, e['dtype']))
            sid = e['shard_id']
            if sid not in files:
                files[sid] = open('%s.data-%05d-of-%05d' % (prefix, sid, shards), 'rb')
            f = files[sid]
            f.seek(e['offset'])
            raw = f.read(e['size'])
            n = int(np.prod(e['shape'])) if e['shape'] else 1
            if len(raw) != e['size'] or e['size'] != n * np.dtype(dt).itemsize:
                raise ValueError('%s: %d bytes on disk, shape %s of %s needs %d' % (k, len(raw), e['shape'], np.dtype(dt).name,
                                                                                    n * np.dtype(dt).itemsize))
            if verify_tensors and e['crc32c'] is not None and mask_crc(crc32c(raw)) != e['crc32c']:
                raise ValueError('%s: tensor CRC mismatch' % k)
            out[k] = np.frombuffer(raw, dtype=np.dtype(dt).newbyteorder('<')).reshape(e['shape']).copy()
    finally:
        for f in files.values():
            f.close()
    return out

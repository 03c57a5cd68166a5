"""Test-only WRITER of the tensor-bundle checkpoint format (`.index` LevelDB-style table + `.data-00000-of-00001`), written
independently of tacotron_amd/tf_bundle.py's reader from the same public format descriptions: sorted keys, prefix compression with
a restart point every `restart_interval` entries, data blocks cut at `block_size` bytes, an index block of (last key of block ->
handle) pairs, an empty metaindex block, the 48-byte footer; every block followed by its compression byte and masked CRC-32C.
`snappy=True` stores blocks in snappy's raw format (literal-only elements: valid streams that exercise the decoder's framing).
TensorFlow is not available here, so files produced by this writer are what the reader is exercised against (tests/test_host.py)."""
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}


def _tab():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_T = _tab()


def crc32c(b):
    c = 0xFFFFFFFF
    for x in b:
        c = _T[(c ^ x) & 255] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def vint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def field(num, wt, payload):
    return vint((num << 3) | wt) + payload


def snappy_literals(b):
    out = bytearray(vint(len(b)))
    pos = 0
    while pos < len(b):
        chunk = b[pos:pos + 3000]
        n = len(chunk) - 1
        if n < 60:
            out.append(n << 2)
        elif n < 256:
            out += bytes([60 << 2, n])
        else:
            out += bytes([61 << 2]) + struct.pack('<H', n)
        out += chunk
        pos += len(chunk)
    return bytes(out)


class _Block:
    def __init__(self, restart_interval):
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b''
        self.ri = restart_interval

    def add(self, key, value):
        shared = 0
        if self.count and self.count % self.ri == 0:
            self.restarts.append(len(self.buf))
        elif self.count:
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
        self.buf += vint(shared) + vint(len(key) - shared) + vint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def _emit(f, body, snappy):
    ctype = 1 if snappy else 0
    payload = snappy_literals(body) if snappy else body
    off = f.tell()
    trailer = bytes([ctype])
    f.write(payload + trailer + struct.pack('<I', masked(crc32c(payload + trailer))))
    return vint(off) + vint(len(payload))


def write_table(path, items, block_size=4096, restart_interval=16, snappy=False):
    items = sorted(items)
    with open(path, 'wb') as f:
        index = _Block(1)
        blk = _Block(restart_interval)
        for k, v in items:
            blk.add(k, v)
            if len(blk.buf) >= block_size:
                index.add(blk.last, _emit(f, blk.finish(), snappy))
                blk = _Block(restart_interval)
        if blk.count:
            index.add(blk.last, _emit(f, blk.finish(), snappy))
        meta = _emit(f, _Block(1).finish(), False)
        idx = _emit(f, index.finish(), snappy)
        footer = meta + idx
        f.write(footer + b'\0' * (40 - len(footer)) + struct.pack('<Q', MAGIC))


def write_bundle(prefix, variables, block_size=4096, snappy=False):
    """variables: {name: numpy array}."""
    items = [(b'', field(1, 0, vint(1)) + field(2, 0, vint(0)) + field(3, 2, vint(2) + field(1, 0, vint(1))))]
    off = 0
    with open(prefix + '.data-00000-of-00001', 'wb') as d:
        for name in sorted(variables):
            a = np.asarray(variables[name])
            a = a if a.flags['C_CONTIGUOUS'] else np.array(a, order='C')
            raw = a.astype(a.dtype.newbyteorder('<')).tobytes()
            shape = b''.join(field(2, 2, (lambda p: vint(len(p)) + p)(field(1, 0, vint(int(s))))) for s in a.shape)
            e = field(1, 0, vint(DT[a.dtype])) + field(2, 2, vint(len(shape)) + shape)
            if off:
                e += field(4, 0, vint(off))
            e += field(5, 0, vint(len(raw))) + field(6, 5, struct.pack('<I', masked(crc32c(raw))))
            items.append((name.encode(), e))
            d.write(raw)
            off += len(raw)
    write_table(prefix + '.index', items, block_size=block_size, snappy=snappy)

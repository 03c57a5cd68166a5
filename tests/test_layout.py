"""CPU: the C-ABI library loads, exports every symbol include/taco_hip.h declares, and its parameter table equals
the oracle's TF-order spec.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import taco_numpy as on

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header(built_lib):
    hdr = open(os.path.join(ROOT, 'include', 'taco_hip.h')).read()
    declared = set(re.findall(r'\b(taco_[a-z_0-9]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    raw = ctypes.CDLL(built_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), 'libtaco_hip.so does not export %s' % name
    assert declared == set(built_lib.EXPORTS), (declared ^ set(built_lib.EXPORTS))
    declared_v = int(re.search(r'#define\s+TACO_VERSION\s+(\d+)', hdr).group(1))
    assert built_lib.version() == declared_v


@pytest.mark.parametrize('V,r,S', [(60, 2, 1), (20, 5, 1), (33, 3, 1), (60, 2, 109), (20, 2, 7)])
def test_param_table_matches_oracle_spec(built_lib, V, r, S):
    shape = built_lib.make_shape(4, 10, 6, r, V, S)
    table = built_lib.param_table(shape)
    spec = on.param_spec(V, r, S)
    assert len(table) == len(spec)
    off = 0
    for (name, o, size, dims), (n2, shp, _) in zip(table, spec):
        assert name == n2 and o == off and tuple(shp) == dims and size == int(np.prod(shp))
        assert o % 4 == 0, '%s not 16-byte aligned' % name
        off += size
    assert built_lib.param_count(shape) == off


def test_nancy_param_count(built_lib):
    assert built_lib.param_count(built_lib.make_shape(32, 200, 180, 2, 60)) == 6926609
    # VCTK config 5: 109 speakers (speaker table + 4 x (spk dense + 256->128 adapter) + GRU-init dense)
    assert built_lib.param_count(built_lib.make_shape(32, 200, 180, 2, 60, 109)) == 7070817


def test_workspace_table(built_lib):
    shape = built_lib.make_shape(32, 200, 180, 2, 60)
    for train in (True, False):
        rows = built_lib.workspace_table(shape, train)
        total = built_lib.workspace_bytes(shape, train)
        names = [r[0] for r in rows]
        assert 'enc.bank' in names and 'post.out' in names and ('dec.stash' in names) == train
        end = 0
        for name, off, size, dims in rows:
            assert off % 64 == 0 and off >= end, name
            end = off + size
        assert end * 4 <= total
    assert built_lib.workspace_bytes(shape, True) > built_lib.workspace_bytes(shape, False)


def test_error_behaviour(built_lib):
    bad = built_lib.make_shape(0, 10, 6, 2, 60)
    with pytest.raises(built_lib.TacoError):
        built_lib.param_count(bad)
    assert 'B=0' in built_lib.last_error()
    with pytest.raises(built_lib.TacoError):
        built_lib.workspace_bytes(built_lib.make_shape(2, 10, 6, 9, 60), True)
    assert 'r=9' in built_lib.last_error()


@pytest.mark.parametrize('S', [1, 109])
def test_gradient_segments_cover_the_buffer(built_lib, S):
    """taco_grad_segments (include/taco_hip.h, version 118): five contiguous segments that tile the flat gradient buffer; the
    encoder conv bank is a segment of its own (K = 16 widths: 136 taps x 128 x 128 kernels + 16 biases + BN gamma / beta), and
    every segment has a name in the reducer's report."""
    from tacotron_amd.dist import SEGMENT_NAMES
    shape = built_lib.make_shape(32, 200, 180, 2, 60, S)
    b = built_lib.grad_segments(shape)
    assert len(b) == 6 and b[0] == 0 and b[-1] == built_lib.param_count(shape)
    assert all(b[i] < b[i + 1] for i in range(5))
    assert set(SEGMENT_NAMES) == set(range(5))
    table = {name: (off, size) for name, off, size, _ in built_lib.param_table(shape)}
    assert b[1] == table['encoder/cbhg/bank_1/kernel'][0] and b[2] == table['encoder/cbhg/proj1/kernel'][0]
    assert b[2] - b[1] == 136 * 128 * 128 + 16 * 128 + 2 * 16 * 128
    assert b[4] == table['post/cbhg/bank_1/kernel'][0]

"""CPU-side checks (no GPU): C-ABI exports, host codecs against the oracle, file formats, checkpoint layout."""
import os
import re
import numpy as np
import pytest
import torch

from oracle import pcgc_oracle as orc
from pcgcv2_amd import ops, synthetic, PcgcError
from pcgcv2_amd._lib import lib, SIGNATURES, LIB_PATH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'pcgc_hip.h')).read()
    body = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(pcgc_\w+)\s*\(', body))
    assert len(declared) >= 30
    l = lib()
    for name in declared:
        assert hasattr(l, name), f'{name} declared in include/pcgc_hip.h but not exported by {LIB_PATH}'
    assert declared == set(SIGNATURES), declared ^ set(SIGNATURES)
    assert l.pcgc_version() >= 1


def test_reftable_library_exports_its_header():
    from pcgcv2_amd._lib import reftable_lib, REFTABLE_PATH
    header = open(os.path.join(ROOT, 'include', 'pcgc_reftable.h')).read()
    declared = set(re.findall(r'\b(pcgc_\w+)\s*\(', re.sub(r'/\*.*?\*/', '', header, flags=re.S)))
    assert declared == {'pcgc_reference_table', 'pcgc_reference_table_clear'}
    for name in declared:
        assert hasattr(reftable_lib(), name), f'{name} not exported by {REFTABLE_PATH}'


@pytest.fixture(params=[0, 1], ids=['rc-auto', 'rc-scalar'])
def rc_impl(request):
    ops.set_rc_impl(request.param)
    yield request.param
    ops.set_rc_impl(0)


def test_range_coder_matches_oracle_bit_for_bit(golden_dir, rc_impl):
    rng = np.random.default_rng(1)
    g = np.load(os.path.join(golden_dir, 'entropy_tables.npz'))
    for case, (lo, hi) in [('c1', (-20, 20)), ('c0', (-8, 9)), ('c3', (0, 0)), ('c2', (-3, 2))]:
        table = orc.cdf_u16(orc.cdf_float(g[f'{case}_params'], lo, hi))
        L = hi - lo + 1
        for n in (1, 3, 257, 5000):
            # peaked symbol distribution like real latents
            sym = np.clip(np.rint(rng.normal(L / 2, max(L / 8, 0.3), size=(n, 8))), 0, L - 1).astype(np.int16)
            a = ops.rc_encode(table, sym)
            b = orc.rc_encode(table, sym)
            assert a == b
            np.testing.assert_array_equal(ops.rc_decode(table, a, sym.size), sym.ravel())
            np.testing.assert_array_equal(orc.rc_decode(table, a, sym.size), sym.ravel())


def _table_from_pmf(pmf):
    """[C, L] probabilities -> torchac-normalised uint16 table [C, L+1] through the oracle's quantiser."""
    cdf = np.concatenate([np.zeros((len(pmf), 1)), np.cumsum(pmf / pmf.sum(1, keepdims=True), 1)], 1).clip(0, 1)
    return orc.cdf_u16(cdf.astype(np.float32))


@pytest.mark.parametrize('shape', ['two_symbols_skewed', 'half_half', 'uniform21', 'tails63', 'wide100', 'wide300', 'spike_mid'])
def test_range_coder_adversarial_tables(shape, rc_impl):
    """Tables that stress each renormalisation case against the bit-serial oracle: near-certain symbols (no shared bits
    for many symbols), p = 1/2 around the interval midpoint (long E3 pending chains), minimum-probability symbols
    (17-bit shifts), alphabets on both sides of the 64-boundary SIMD limit."""
    rng = np.random.default_rng(11)
    C = 8
    if shape == 'two_symbols_skewed':
        pmf = np.tile([0.999, 0.001], (C, 1)); draw = lambda n: (rng.random((n, C)) < 0.002).astype(np.int16)
    elif shape == 'half_half':
        pmf = np.tile([0.5, 0.5], (C, 1)); draw = lambda n: np.tile(np.array([0, 1, 1, 0, 1, 0, 0, 1], np.int16), (n, 1)) ^ (rng.random((n, C)) < 0.02)
    elif shape == 'uniform21':
        pmf = np.ones((C, 21)); draw = lambda n: rng.integers(0, 21, (n, C)).astype(np.int16)
    elif shape == 'tails63':
        x = np.arange(63) - 31; pmf = np.tile(np.exp(-0.5 * (x / 1.5) ** 2) + 1e-12, (C, 1)); draw = lambda n: rng.integers(0, 63, (n, C)).astype(np.int16)
    elif shape == 'wide100':
        x = np.arange(100) - 50; pmf = np.tile(np.exp(-np.abs(x) / 9.0), (C, 1)); draw = lambda n: np.clip(np.rint(rng.laplace(50, 9, (n, C))), 0, 99).astype(np.int16)
    elif shape == 'wide300':
        x = np.arange(300) - 150; pmf = np.tile(np.exp(-0.5 * (x / 40.0) ** 2) + 1e-9, (C, 1)); draw = lambda n: np.clip(np.rint(rng.normal(150, 40, (n, C))), 0, 299).astype(np.int16)
    else:
        pmf = np.tile([1e-6, 0.25, 0.5 - 2e-6, 0.25, 1e-6], (C, 1)); draw = lambda n: rng.choice(5, (n, C), p=[0.02, 0.24, 0.48, 0.24, 0.02]).astype(np.int16)
    table = _table_from_pmf(np.asarray(pmf, np.float64))
    for n in (1, 2, 9, 1000, 20000):
        sym = np.ascontiguousarray(draw(n).astype(np.int16))
        a = ops.rc_encode(table, sym)
        if n <= 1000:
            assert a == orc.rc_encode(table, sym)
            np.testing.assert_array_equal(orc.rc_decode(table, a, sym.size), sym.ravel())
        np.testing.assert_array_equal(ops.rc_decode(table, a, sym.size), sym.ravel())
        # decoding index: same stream; segments decoded from the recorded decoder states give the same symbols (E3 tails, p = 1/2
        # chains and no-shift symbols all cross segment boundaries in these tables)
        for k in (1, 3, 8):
            a2, index = ops.rc_encode(table, sym, checkpoints=k)
            assert a2 == a
            for threads in (1, 3):
                ops.set_rc_threads(threads)
                np.testing.assert_array_equal(ops.rc_decode(table, a, sym.size, index=index), sym.ravel())
            ops.set_rc_threads(0)
    # a truncated / corrupt stream must not crash either decoder
    ops.rc_decode(table, a[:len(a) // 2], sym.size)
    ops.rc_decode(table, bytes(rng.integers(0, 256, 64, dtype=np.uint8)), 4096)
    with pytest.raises(PcgcError):                                   # (its index points beyond the truncated stream: refused)
        ops.rc_decode(table, a[:len(a) // 2], sym.size, index=index)


def test_range_coder_long_pending_runs_and_exact_buffers(rc_impl):
    """The encoder's rare paths against the bit-serial oracle: (1) a symbol of probability 2^-15 that straddles the interval midpoint,
    repeated — every one defers ~15 E3 bits, so the pending count passes the 56 bits the one-field emission holds (4 repeats) and the
    32 bits of one run chunk (put_run); (2) an output buffer of exactly the stream's size (the speculative 8-byte stores of the bit
    writer must not be what puts the last bytes in place), and one byte less (refused with the size needed)."""
    C = 8
    pmf = np.tile([0.5 - 2.0 ** -16, 2.0 ** -15, 0.5 - 2.0 ** -16], (C, 1))
    table = _table_from_pmf(np.asarray(pmf, np.float64))
    rng = np.random.default_rng(3)
    for n, p_mid in ((2, 1.0), (40, 1.0), (125, 0.9), (1000, 0.7)):
        sym = np.where(rng.random((n, C)) < p_mid, 1, rng.integers(0, 3, (n, C))).astype(np.int16)
        a = ops.rc_encode(table, sym)
        assert a == orc.rc_encode(table, sym)
        np.testing.assert_array_equal(ops.rc_decode(table, a, sym.size), sym.ravel())
        a2, index = ops.rc_encode(table, sym, checkpoints=4)
        assert a2 == a
        np.testing.assert_array_equal(ops.rc_decode(table, a, sym.size, index=index), sym.ravel())
        cdf = np.ascontiguousarray(table, np.uint16)
        flat = np.ascontiguousarray(sym.ravel())
        for slack in (0, 1, 7, 8):
            buf = np.full(len(a) + slack + 16, 0xA5, np.uint8)                     # 16 guard bytes behind the capacity
            got = int(ops.lib().pcgc_rc_encode(cdf.ctypes.data, C, cdf.shape[1], flat.ctypes.data, flat.size, buf.ctypes.data, len(a) + slack))
            assert got == len(a) and buf[:got].tobytes() == a
            assert (buf[len(a) + slack:] == 0xA5).all(), 'wrote past the capacity'
        buf = np.full(len(a) + 16, 0xA5, np.uint8)
        assert int(ops.lib().pcgc_rc_encode(cdf.ctypes.data, C, cdf.shape[1], flat.ctypes.data, flat.size, buf.ctypes.data, len(a) - 1)) == -len(a)
        assert (buf[len(a) - 1:] == 0xA5).all()


def test_library_crc32_is_zlibs():
    """pcgc_crc32 (carry-less-multiply fold of the sidecar's stream CRC) against zlib.crc32: every length around the 16 / 64-byte
    block boundaries and the 256-byte switch-over, running values, unaligned starts."""
    import zlib
    rng = np.random.default_rng(9)
    blob = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()
    for n in list(range(0, 400)) + [1023, 1024, 4097, 65535, 69999]:
        for off in (0, 1, 13):
            assert ops.crc32(blob[off:off + n]) == zlib.crc32(blob[off:off + n]), (n, off)
    run_a = run_b = 0
    for lo, hi in ((0, 5), (5, 300), (300, 301), (301, 40000), (40000, 70000)):
        run_a, run_b = ops.crc32(blob[lo:hi], run_a), zlib.crc32(blob[lo:hi], run_b)
        assert run_a == run_b
    assert run_a == zlib.crc32(blob)


def test_range_decoder_rejects_malformed_index():
    rng = np.random.default_rng(5)
    table = _table_from_pmf(np.tile(np.exp(-0.5 * ((np.arange(21) - 10) / 2.0) ** 2), (8, 1)))
    sym = np.clip(np.rint(rng.normal(10, 2, (4000, 8))), 0, 20).astype(np.int16)
    a, index = ops.rc_encode(table, sym, checkpoints=4)
    np.testing.assert_array_equal(ops.rc_decode(table, a, sym.size, index=index), sym.ravel())
    for breakage in ('order', 'range', 'phase', 'start', 'bitpos'):
        bad = index.copy()
        if breakage == 'order':
            bad[[1, 2]] = bad[[2, 1]]
        elif breakage == 'range':
            bad[3, 0] = sym.size + 8
        elif breakage == 'phase':
            bad[2, 0] += 1                                   # not a row boundary
        elif breakage == 'start':
            bad[0, 3] = 7                                    # the first segment must start from the initial coder state
        else:
            bad[1, 1] = 8 * len(a) + 1000                    # bit position beyond the stream
        with pytest.raises(PcgcError):
            ops.rc_decode(table, a, sym.size, index=bad)


def test_entropy_pools_under_concurrent_callers():
    """Frames in flight: several host threads use the indexed range decoder and the grouped coordinate codec at the same time
    (each pool serialises its callers; the two pools run side by side)."""
    import threading
    table = _table_from_pmf(np.tile(np.exp(-0.5 * ((np.arange(21) - 10) / 2.0) ** 2), (8, 1)))
    pts = np.unique(synthetic.shell('shell10').numpy() // 8, axis=0).astype(np.int32)
    want_pts = ops.oct_decode(ops.oct_encode(pts))
    errors = []

    def worker(seed):
        rng = np.random.default_rng(seed)
        for _ in range(8):
            sym = np.clip(np.rint(rng.normal(10, 2, (5000, 8))), 0, 20).astype(np.int16)
            stream, index = ops.rc_encode(table, sym, checkpoints=8)
            if not np.array_equal(ops.rc_decode(table, stream, sym.size, index=index), sym.ravel()):
                errors.append(('range', seed))
            if not np.array_equal(ops.oct_decode(ops.oct_encode(pts)), want_pts):
                errors.append(('octree', seed))
    ops.set_rc_threads(4)
    try:
        threads = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
        [t.start() for t in threads]
        [t.join() for t in threads]
    finally:
        ops.set_rc_threads(0)
    assert not errors, errors


def test_feature_index_sidecar_is_tied_to_its_stream(tmp_path):
    """coder._pack_index / _load_sidecar: the sidecar is used only for the stream it was written with, and only if it is intact —
    a damaged checkpoint word (header untouched) must not reach the decoder (the body is covered by a CRC)."""
    from pcgcv2_amd import coder
    index = np.arange(2 * ops.RC_CKPT_WORDS, dtype=np.uint32).reshape(2, -1)
    path = str(tmp_path / 'x_F.idx')
    blob = coder._pack_index(b'stream-bytes', index, table_crc=0xDEADBEEF)
    with open(path, 'wb') as fh:
        fh.write(blob)
    got, crc = coder._load_sidecar(path, b'stream-bytes')
    np.testing.assert_array_equal(got, index)
    assert crc == 0xDEADBEEF
    assert coder._load_index(path, b'stream-bytez') is None          # same length, other content
    assert coder._load_index(path, b'stream-bytes+') is None
    assert coder._load_index(str(tmp_path / 'missing.idx'), b'stream-bytes') is None
    for pos in (len(blob) - 3, 17):                                   # a flipped bit in a checkpoint word / in the table-CRC field
        bad = bytearray(blob); bad[pos] ^= 0x10
        with open(path, 'wb') as fh:
            fh.write(bytes(bad))
        assert coder._load_sidecar(path, b'stream-bytes') == (None, None)
    with open(path, 'wb') as fh:
        fh.write(b'PCG2')                                             # truncated
    assert coder._load_index(path, b'stream-bytes') is None
    # a sidecar without checkpoints (short streams) still carries the table guard
    with open(path, 'wb') as fh:
        fh.write(coder._pack_index(b'stream-bytes', None, table_crc=7))
    assert coder._load_sidecar(path, b'stream-bytes') == (None, 7)


def test_range_coder_rejects_out_of_table_symbol():
    table = orc.cdf_u16(np.array([[0, .25, .5, 1.0]] * 8, np.float32))
    with pytest.raises(PcgcError):
        ops.rc_encode(table, np.full((2, 8), 3, np.int16))


@pytest.mark.parametrize('case', ['empty', 'single', 'random', 'shell', 'dense', 'big_coords'])
def test_octree_codec_roundtrip(case):
    rng = np.random.default_rng(5)
    if case == 'empty':
        pts = np.zeros((0, 3), np.int32)
    elif case == 'single':
        pts = np.array([[5, 0, 127]], np.int32)
    elif case == 'random':
        pts = np.unique(rng.integers(0, 128, size=(3000, 3)), axis=0).astype(np.int32)
    elif case == 'dense':
        g = np.arange(8)
        pts = np.stack(np.meshgrid(g, g, g, indexing='ij'), -1).reshape(-1, 3).astype(np.int32)
    elif case == 'big_coords':
        pts = np.unique(rng.integers(0, 1 << 20, size=(500, 3)), axis=0).astype(np.int32)
    else:
        pts = np.unique(synthetic.shell('shell9').numpy() // 8, axis=0).astype(np.int32)          # 4576 stride-8 voxels
    rng.shuffle(pts)
    data = ops.oct_encode(pts)
    back = ops.oct_decode(data)
    assert data[:4] == b'PCGO'
    key = lambda a: a[np.lexsort((a[:, 0], a[:, 1], a[:, 2]))]
    np.testing.assert_array_equal(key(back), key(pts))
    if case == 'shell':
        bits_per_point = 8 * len(data) / len(pts)
        assert bits_per_point < 1.6, bits_per_point              # neighbour contexts from the mixed-shape prior: 1.45 bit/pt here (1.88 from p = 1/2), 1.26 on the vox10 frame's level


def test_octree_stream_bytes_are_pinned():
    """The `_C.bin` container is a FORMAT: files written by one build must decode with the next.  The streams of the bench frame's
    stride-8 level are pinned by hash per version: 2 = one stream / 3 = groups of subtrees (round 3: p = 1/2 / sphere-trained prior; written
    on request, pcgc_set_oct_model(0)), 4 / 5 = the same layouts with context model 1 (round 5: mixed-shape prior + fast start; the
    default).  A faster coder must reproduce them bit for bit; the decoder reads all four."""
    import hashlib
    pts = np.unique(synthetic.shell('shell10').numpy() // 8, axis=0).astype(np.int32)
    small_pts = np.unique(synthetic.shell('shell9').numpy() // 8, axis=0).astype(np.int32)      # below 8192 points: always one stream
    want = {(0, 0): (3434, 2, '7ff1c2c56af7cd67446d0d0aae2b0752effeb22291bac33a3fc289364b1bfb35'),
            (0, 1): (3949, 3, '43d8e8e41bcdbf50c0573deb0f7b228ae004a9bed2e1c9bba9f25e2f237870db'),
            (1, 0): (2952, 4, 'bbc009f395b7744422b16705f51788445e42d190b7ef2b77936e712173d8f118'),
            (1, 1): (3580, 5, '11a722af42a253f990abeca36aba826df0df00836aadc06d57ecda01a88d1328')}
    small_want = {0: (1076, 2, '9e803865eb61dcd08fb477e287cf146577effc8ebbc111a9ddbed875aeb5ae64'),
                  1: (830, 4, 'b0d8cbb443d365a693e92fc08cbddeef654e41992c5f790d27f988a06ebaf95e')}
    key = lambda a: a[np.lexsort((a[:, 0], a[:, 1], a[:, 2]))]
    streams = []
    try:
        for (model, tiled), (size, version, digest) in want.items():
            ops.set_oct_model(model)
            ops.set_oct_tiled(tiled)
            data = ops.oct_encode(pts)
            assert (len(data), data[4]) == (size, version)
            assert hashlib.sha256(data).hexdigest() == digest
            streams.append(data)
        for model, (size, version, digest) in small_want.items():
            ops.set_oct_model(model)
            ops.set_oct_tiled(1)
            small = ops.oct_encode(small_pts)
            assert (len(small), small[4]) == (size, version)
            assert hashlib.sha256(small).hexdigest() == digest
            np.testing.assert_array_equal(key(ops.oct_decode(small)), key(small_pts))
    finally:
        ops.set_oct_tiled(1)
        ops.set_oct_model(1)
    for data in streams:                                         # whatever the encoder is set to, the decoder reads every version
        np.testing.assert_array_equal(key(ops.oct_decode(data)), key(pts))
    assert 8 * len(streams[3]) / len(pts) < 1.55 and 8 * len(streams[2]) / len(pts) < 1.30      # model 1: 1.53 (groups) / 1.26 (one stream) bit per point


@pytest.mark.parametrize('groups', [0, 1, 3, 12])
def test_octree_codec_groups_of_subtrees(groups):
    """Stream version 3 (independent groups of subtrees, coded and decoded side by side) against version 2 on the stride-8 level
    of the vox10 bench frame: same voxels, in the same (Morton) order, for every group count and thread count."""
    pts = np.unique(synthetic.shell('shell10').numpy() // 8, axis=0).astype(np.int32)              # 18 732 voxels
    try:
        for model in (0, 1):                                       # the round-3 streams (versions 2 / 3) and the round-5 ones (4 / 5)
            ops.set_oct_model(model)
            ops.set_oct_tiled(0)
            plain = ops.oct_encode(pts)
            ops.set_oct_tiled(groups)
            data = ops.oct_encode(pts)
            assert data[4] == ((2 if groups == 0 else 3) if model == 0 else (4 if groups == 0 else 5))
            for threads in (1, 4):
                ops.set_rc_threads(threads)
                np.testing.assert_array_equal(ops.oct_decode(data), ops.oct_decode(plain))
            assert 8 * len(data) / len(pts) < (1.5 if groups == 0 else 2.0)
            if groups:                                             # damage in one group is detected, not decoded into garbage silently
                bad = bytearray(data); bad[len(bad) // 2] ^= 0x55; bad[len(bad) // 2 + 1] ^= 0xAA
                try:
                    back = ops.oct_decode(bytes(bad))
                    assert len(back) == len(pts)
                except PcgcError:
                    pass
    finally:
        ops.set_oct_tiled(1); ops.set_oct_tiled(8); ops.set_rc_threads(0); ops.set_oct_model(1)


def test_octree_rejects_foreign_stream():
    with pytest.raises(PcgcError):
        ops.oct_decode(b'not a stream at all')


def test_frame_decode_from_several_threads(tmp_path):
    """Four host threads decode four different clouds at once (shard.code_units(in_flight=F) does this with one Coder per thread): the
    pools are shared (one run at a time each, helpers woken ahead), the scratch buffers are per thread — every result must be the
    cloud's own."""
    import threading
    from pcgcv2_amd.entropy_model import EntropyBottleneck
    torch.manual_seed(9)
    eb = EntropyBottleneck(8)
    packed = eb._host_packed()
    clouds = []
    for t in range(4):
        rng = np.random.default_rng(100 + t)
        r = 3000 + 1500 * t
        sym = np.clip(np.rint(rng.normal(6 + t, 2.0, size=(r, 8))), 0, 14 + t).astype(np.int16)
        sym[0, 0], sym[-1, -1] = 0, 14 + t
        xyz = rng.permutation(np.unique(rng.integers(0, 70 + 10 * t, size=(4 * r, 3)), axis=0))[:r].astype(np.int32)
        stem = str(tmp_path / f'c{t}')
        ops.items_encode([stem], sym, xyz, [r], [(-7.0 - t, 7.0)], [(r, 2 * r, 3 * r)], packed, 16)
        want = xyz[np.lexsort((xyz[:, 0], xyz[:, 1], xyz[:, 2]))] * 8
        clouds.append((stem, r, sym, want))
    errors = []
    def worker(t):
        stem, r, sym, want = clouds[t]
        sym_buf, level_buf = np.zeros((r, 8), np.int16), np.zeros((r, 4), np.int32)
        for _ in range(12):
            sym_buf[:] = -1; level_buf[:] = -1
            n, rng_, counts, native = ops.frame_decode(stem, 8, packed, sym_buf, level_buf)
            if n != r or not native or counts != (r, 2 * r, 3 * r) or not np.array_equal(sym_buf, sym) or not np.array_equal(level_buf[:, 1:], want) \
                    or level_buf[:, 0].any():
                errors.append(t)
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors


def test_frame_decode_survives_damaged_files(tmp_path):
    """pcgc_frame_decode on truncated / bit-flipped files: an error (PcgcError) or a decode — never a crash, a hang or a write outside the
    caller's buffers (guard rows behind the capacity stay untouched)."""
    from pcgcv2_amd.entropy_model import EntropyBottleneck
    rng = np.random.default_rng(77)
    torch.manual_seed(5)
    eb = EntropyBottleneck(8)
    r = 6000
    sym = np.clip(np.rint(rng.normal(8, 2.5, size=(r, 8))), 0, 16).astype(np.int16)
    sym[0, 0], sym[-1, -1] = 0, 16
    xyz = rng.permutation(np.unique(rng.integers(0, 90, size=(4 * r, 3)), axis=0))[:r].astype(np.int32)
    stem = str(tmp_path / 'f')
    ops.items_encode([stem], sym, xyz, [r], [(-8.0, 8.0)], [(r * 3, r * 9, r * 30)], eb._host_packed(), 16)
    good = {sfx: open(stem + sfx, 'rb').read() for sfx in ('_C.bin', '_F.bin', '_H.bin', '_num_points.bin', '_F.idx')}
    def attempt():
        sym_buf, level_buf = np.full((r + 8, 8), -3, np.int16), np.full((r + 8, 4), -3, np.int32)
        try:
            ops.frame_decode(stem, 8, eb._host_packed(), sym_buf[:r + 4], level_buf[:r + 4])
        except PcgcError:
            pass
        assert (sym_buf[r + 4:] == -3).all() and (level_buf[r + 4:] == -3).all()
    attempt()
    for sfx, blob in good.items():
        for kind in ('truncate_half', 'truncate_3', 'flip_early', 'flip_late', 'empty'):
            bad = bytearray(blob)
            if kind == 'truncate_half': bad = bad[:len(bad) // 2]
            elif kind == 'truncate_3': bad = bad[:max(0, len(bad) - 3)]
            elif kind == 'flip_early' and len(bad) > 6: bad[5] ^= 0x5A
            elif kind == 'flip_late' and len(bad) > 6: bad[len(bad) * 3 // 4] ^= 0xFF
            elif kind == 'empty': bad = bytearray()
            with open(stem + sfx, 'wb') as fh:
                fh.write(bytes(bad))
            attempt()
        with open(stem + sfx, 'wb') as fh:
            fh.write(blob)
    n, rng_, counts, native = ops.frame_decode(stem, 8, eb._host_packed(), np.zeros((r, 8), np.int16), np.zeros((r, 4), np.int32))
    assert (n, native, counts) == (r, True, (r * 3, r * 9, r * 30))
    # a header whose range is not integral (what a flipped mantissa bit makes of it): the table callback would size its output from
    # arange(min_v, max_v + 1) — one entry more than (int)(max_v - min_v) + 2 — so the range is refused before any table is built
    import struct
    head = bytearray(good['_H.bin'])
    head[9:13] = struct.pack('<f', -8.5)
    with open(stem + '_H.bin', 'wb') as fh:
        fh.write(bytes(head))
    with pytest.raises(PcgcError, match='symbol range'):
        ops.frame_decode(stem, 8, eb._host_packed(), np.zeros((r, 8), np.int16), np.zeros((r, 4), np.int32))
    with pytest.raises(PcgcError):
        eb.reference_table_native(np.float32(-8.5), np.float32(8.0))
    head = bytearray(good['_H.bin'])
    head[0:4] = struct.pack('<i', 1 << 30)                     # a row count no cloud has: refused, not handed to the caller as a size to allocate
    with open(stem + '_H.bin', 'wb') as fh:
        fh.write(bytes(head))
    with pytest.raises(PcgcError, match='implausible'):
        ops.frame_decode(stem, 8, eb._host_packed(), np.zeros((r, 8), np.int16), np.zeros((r, 4), np.int32))


def test_state_dict_layout_matches_reference(golden_dir):
    from pcgcv2_amd.pcc_model import PCCModel
    m = PCCModel()
    sd = m.state_dict()
    assert len(sd) == 227                                       # 106 convs x 2 + 15 entropy tensors
    want = {}
    for line in open(os.path.join(golden_dir, 'state_dict_keys.txt')):
        k, shape = line.split(' ', 1)
        want['entropy_bottleneck.' + k] = eval(shape)
    for k, shape in want.items():
        assert list(sd[k].shape) == shape, k
    assert list(sd['encoder.conv0.kernel'].shape) == [27, 1, 16]
    assert list(sd['encoder.conv0.bias'].shape) == [1, 16]
    assert list(sd['encoder.down0.kernel'].shape) == [8, 16, 32]
    assert list(sd['encoder.block0.1.conv1_0.kernel'].shape) == [32, 8]        # k=1 kernels are 2-D in ME
    assert list(sd['decoder.up0.kernel'].shape) == [8, 8, 64]
    assert list(sd['decoder.conv2_cls.kernel'].shape) == [27, 16, 1]
    assert sum(p.numel() for p in m.encoder.parameters()) == 408008
    assert sum(p.numel() for p in m.decoder.parameters()) == 370127
    # a checkpoint in the reference's layout loads strictly, aliases included
    syn = synthetic.synthetic_state_dict()
    m.load_state_dict(syn, strict=True)
    assert m.entropy_bottleneck.matrix is m.entropy_bottleneck._matrices[3]
    np.testing.assert_array_equal(orc.pack_eb_params(synthetic.state_dict_to_numpy(syn)),
                                  m.entropy_bottleneck.packed_params(torch.device('cpu')).numpy())


def test_ply_io_matches_reference_format(golden_dir, tmp_path):
    from pcgcv2_amd.data_utils import read_ply_ascii_geo, write_ply_ascii_geo
    g = np.load(os.path.join(golden_dir, 'ply_format.npz'))
    p = tmp_path / 'w.ply'
    write_ply_ascii_geo(str(p), g['coords'])
    assert p.read_bytes() == g['file_bytes'].tobytes()
    np.testing.assert_array_equal(read_ply_ascii_geo(str(p)), g['read_back'])
    p2 = tmp_path / 'b.ply'
    p2.write_bytes(g['file2_bytes'].tobytes())
    np.testing.assert_array_equal(read_ply_ascii_geo(str(p2)), g['read_back2'])


def test_native_d1_matches_pc_error_d(golden_dir, tmp_path):
    from pcgcv2_amd.pc_error import d1_psnr, pc_error
    from pcgcv2_amd.data_utils import write_ply_ascii_geo
    g = np.load(os.path.join(golden_dir, 'd1_metric.npz'))
    for i in range(int(g['n_cases'])):
        m = d1_psnr(g[f'p{i}_a'], g[f'p{i}_b'], int(g[f'p{i}_res']))
        assert m['mseF      (p2point)'] == pytest.approx(float(g[f'p{i}_mseF(p2point)']), rel=1e-5, abs=1e-9)
        assert m['h.        (p2point)'] == pytest.approx(float(g[f'p{i}_h.(p2point)']), rel=1e-5, abs=1e-9)
        if m['mseF      (p2point)'] > 0:
            assert m['mseF,PSNR (p2point)'] == pytest.approx(float(g[f'p{i}_mseF_PSNR(p2point)']), abs=2e-4)
    a, b = tmp_path / 'a.ply', tmp_path / 'b.ply'
    write_ply_ascii_geo(str(a), g['p0_a']); write_ply_ascii_geo(str(b), g['p0_b'])
    df = pc_error(str(a), str(b), res=int(g['p0_res']))
    assert df['mseF,PSNR (p2point)'][0] == pytest.approx(float(g['p0_mseF_PSNR(p2point)']), abs=2e-4)


def test_native_d2_matches_pc_error_d(golden_dir, tmp_path, monkeypatch):
    """point-to-plane (the reference's test.py:74-75 asks pc_error for it with normal=True): the native host computation against the vendored
    binary's stdout (golden G6: four cloud pairs with normals, distance ties, identical clouds, scattered points) — every column the reference
    parses, to the six digits the tool prints; then through pc_error() itself on PLY files with and without normals"""
    from pcgcv2_amd import pc_error as pe
    from pcgcv2_amd.data_utils import write_ply_ascii_geo
    g = np.load(os.path.join(golden_dir, 'd2_metric.npz'))
    cols = ['mse1      (p2point)', 'mse2      (p2point)', 'mseF      (p2point)', 'mse1      (p2plane)', 'mse2      (p2plane)', 'mseF      (p2plane)',
            'h.       1(p2point)', 'h.       2(p2point)', 'h.        (p2point)']
    psnr = ['mse1,PSNR (p2point)', 'mse2,PSNR (p2point)', 'mseF,PSNR (p2point)', 'mse1,PSNR (p2plane)', 'mse2,PSNR (p2plane)', 'mseF,PSNR (p2plane)']
    gk = lambda i, key: float(g[f'p{i}_' + key.replace(' ', '').replace(',', '_')])
    for i in range(int(g['n_cases'])):
        m = pe.d2_psnr(g[f'p{i}_a'], g[f'p{i}_na'], g[f'p{i}_b'], int(g[f'p{i}_res']))
        for key in cols:
            assert m[key] == pytest.approx(gk(i, key), rel=2e-5, abs=1e-9), (i, key)
        for key in psnr:
            if np.isinf(gk(i, key)):
                assert np.isinf(m[key])
            else:
                assert m[key] == pytest.approx(gk(i, key), abs=2e-4), (i, key)
    monkeypatch.setattr(pe, '_exe', lambda: None)                      # (no binary: the native path)
    a, b = tmp_path / 'a.ply', tmp_path / 'b.ply'
    with open(a, 'w') as f:
        f.write('ply\nformat ascii 1.0\ncomment normals from a mesh\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n'
                'property float nx\nproperty float ny\nproperty float nz\nend_header\n' % len(g['p0_a']))
        for q, n in zip(g['p0_a'], g['p0_na']):
            f.write('%d %d %d %.6f %.6f %.6f\n' % (q[0], q[1], q[2], n[0], n[1], n[2]))
    write_ply_ascii_geo(str(b), g['p0_b'])
    df = pe.pc_error(str(a), str(b), res=int(g['p0_res']), normal=True)
    assert df['mseF,PSNR (p2plane)'][0] == pytest.approx(gk(0, 'mseF,PSNR (p2plane)'), abs=2e-4)
    assert df['mseF,PSNR (p2point)'][0] == pytest.approx(gk(0, 'mseF,PSNR (p2point)'), abs=2e-4)
    with pytest.raises(ValueError, match='normals'):
        pe.pc_error(str(b), str(a), res=64, normal=True)                # infile1 without normals: as `pc_error_d -n` would fail


def test_product_refuses_cpu_tensors():
    from pcgcv2_amd.sparse import SparseTensor
    with pytest.raises(PcgcError):
        SparseTensor(torch.ones(4, 1), coordinates=torch.zeros(4, 4, dtype=torch.int32), device='cpu')
    with pytest.raises(PcgcError):
        ops.conv_gather(None, torch.ones(4, 4), torch.ones(4, 4), None)


def test_oracle_roundtrip_small_shell():
    """encode -> decode of the oracle itself on shell6: decoded point count follows coder.py:105-112."""
    sd = synthetic.state_dict_to_numpy(synthetic.synthetic_state_dict())
    c = synthetic.shell('shell6').numpy()
    c4 = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
    enc = orc.encode(sd, c4)
    assert len(enc['H']) == 17 and len(enc['num_points']) == 12
    n4, n2, n1 = np.frombuffer(enc['num_points'], np.int32)
    assert n1 == len(c4) and n4 < n2 < n1
    out = orc.decode(sd, enc['coords8'], enc['F'], enc['H'], enc['num_points'])
    assert len(out) == n1 and len(np.unique(out, axis=0)) == n1
    out_half = orc.decode(sd, enc['coords8'][::-1], enc['F'], enc['H'], enc['num_points'], rho=0.5)
    assert len(out_half) == int(0.5 * n1)
    # latent symbols use a realistic alphabet with the synthetic gain
    hdr_min, hdr_max = np.frombuffer(enc['H'][9:17], np.float32)
    assert 4 <= hdr_max - hdr_min <= 200


def test_native_ply_reader_matches_reference_rule(tmp_path):
    """The native reader against the oracle's restatement of the reference loop on awkward inputs."""
    from pcgcv2_amd.data_utils import read_ply_ascii_geo, write_ply_ascii_geo
    rng = np.random.default_rng(11)
    cases = {
        'plain': 'ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nend_header\n1 2 3\n4 5 6\n',
        'floats_colors': 'ply\nend_header\n1.9 2.1 -3.7 255 0 12\n1e2 2.5e1 3 1 1 1\n',
        'no_header': '7 8 9\n10 11 12\n',
        'crlf': 'ply\r\nend_header\r\n1 2 3\r\n4 5 6\r\n',
        'trailing_space': 'end_header\n1 2 3 \n4 5 6\n',             # "3 " -> tokens [...,'3','\n'] : accepted
        'double_space': 'end_header\n1  2 3\n4 5 6\n',              # empty token -> line rejected by the reference
        'no_final_newline': 'end_header\n1 2 3\n4 5 6',
        'text_line_inside': 'end_header\n1 2 3\ncomment here 5\n4 5 6\n',
        'numeric_header_like': 'element vertex 2\n3 4\n1 2 3\n',    # a 2-number line makes the reference fail too
    }
    for name, text in cases.items():
        p = tmp_path / (name + '.ply')
        p.write_bytes(text.encode())
        try:
            want = orc.read_ply_ascii_geo(str(p))
        except Exception:
            want = None
        if want is None:
            with pytest.raises(Exception):
                read_ply_ascii_geo(str(p))
        else:
            np.testing.assert_array_equal(read_ply_ascii_geo(str(p)), want, err_msg=name)
    big = rng.integers(-5, 4096, size=(20000, 3))
    p = tmp_path / 'big.ply'
    write_ply_ascii_geo(str(p), big)
    assert p.read_bytes() == orc.ply_ascii_bytes(big)
    np.testing.assert_array_equal(read_ply_ascii_geo(str(p)), big)
    with pytest.raises(FileNotFoundError):
        read_ply_ascii_geo(str(tmp_path / 'missing.ply'))


# ------------------------------------------------------------------------------------------------ tmc3 subprocess protocol
_TMC3_STUB = r'''#!/usr/bin/env python3
"""Stand-in for the MPEG tmc3 binary in tests: same command line (gpcc.py:11-21,30-36), `--mode=0` stores the PLY text behind a
magic, `--mode=1` writes it back.  It exercises the subprocess protocol + temp-PLY handling, not G-PCC itself."""
import sys
args = dict(a[2:].split('=', 1) for a in sys.argv[1:] if a.startswith('--') and '=' in a)
with open(args['compressedStreamPath'] + '.argv', 'a') as log:
    log.write(' '.join(sys.argv[1:]) + '\n')
if args['mode'] == '0':
    ply = open(args['uncompressedDataPath'], 'rb').read()
    assert ply.startswith(b'ply\nformat ascii 1.0\n'), 'encoder input must be the ASCII PLY the reference writes'
    open(args['compressedStreamPath'], 'wb').write(b'STUBGPCC' + ply)
elif args['mode'] == '1':
    blob = open(args['compressedStreamPath'], 'rb').read()
    assert blob[:8] == b'STUBGPCC'
    assert args.get('outputBinaryPly') == '0'
    open(args['reconstructedDataPath'], 'wb').write(blob[8:])
else:
    sys.exit(3)
'''


@pytest.fixture
def tmc3_stub(tmp_path, monkeypatch):
    exe = tmp_path / 'tmc3'
    exe.write_text(_TMC3_STUB)
    exe.chmod(0o755)
    monkeypatch.setenv('PCGC_TMC3', str(exe))
    return exe


def test_coordinate_coder_runs_the_tmc3_protocol(tmc3_stub, tmp_path):
    """CoordinateCoder with an installed `tmc3` (coder.py:16-36 + gpcc.py:6-41): temp ASCII PLY -> subprocess with the
    reference's flags -> `_C.bin`; decode: subprocess -> temp PLY -> read_ply_ascii_geo; temp files removed."""
    from pcgcv2_amd.coder import CoordinateCoder
    from pcgcv2_amd import gpcc
    assert gpcc.tmc3_path() == str(tmc3_stub)
    rng = np.random.default_rng(0)
    pts = np.unique(rng.integers(0, 128, size=(500, 3)), axis=0).astype(np.int32)
    out = tmp_path / 'out'
    out.mkdir()
    cc = CoordinateCoder(str(out / 'frame'))
    cc.encode(torch.from_numpy(pts), postfix='_r3')                 # the reference passes a CPU torch tensor (coder.py:89)
    assert (out / 'frame_r3_C.bin').read_bytes()[:8] == b'STUBGPCC'
    assert not gpcc.is_native_stream(str(out / 'frame_r3_C.bin'))
    back = cc.decode(postfix='_r3')
    np.testing.assert_array_equal(back, pts)
    assert back.dtype.kind == 'i'
    calls = (out / 'frame_r3_C.bin.argv').read_text().splitlines()
    assert len(calls) == 2
    for flag in ('--mode=0', '--positionQuantizationScale=1', '--trisoupNodeSizeLog2=0', '--neighbourAvailBoundaryLog2=8',
                 '--intra_pred_max_node_size_log2=6', '--inferredDirectCodingMode=0', '--maxNumQtBtBeforeOt=4'):
        assert flag in calls[0].split(), flag                        # gpcc.py:11-19
    assert '--mode=1' in calls[1].split() and '--outputBinaryPly=0' in calls[1].split()      # gpcc.py:30-35
    assert sorted(p.name for p in out.iterdir()) == ['frame_r3_C.bin', 'frame_r3_C.bin.argv']   # no temp PLY left behind


def test_coordinate_coder_tmc3_concurrent_calls_do_not_share_temp_files(tmc3_stub, tmp_path):
    """Several frames coded concurrently with the same prefix (serving mode): each call owns its temp PLY."""
    from concurrent.futures import ThreadPoolExecutor
    from pcgcv2_amd.coder import CoordinateCoder
    rng = np.random.default_rng(1)
    clouds = [np.unique(rng.integers(0, 64, size=(300 + 50 * i, 3)), axis=0).astype(np.int32) for i in range(8)]

    def one(i):
        cc = CoordinateCoder(str(tmp_path / 'f'))                    # same prefix for every worker, as shard.code_units does
        for _ in range(3):
            cc.encode(clouds[i], postfix=f'_u{i}')
            np.testing.assert_array_equal(cc.decode(postfix=f'_u{i}'), clouds[i])
        return True
    with ThreadPoolExecutor(8) as ex:
        assert all(ex.map(one, range(8)))
    assert not [p for p in tmp_path.iterdir() if p.suffix == '.ply']


def test_tmc3_failure_is_reported(tmp_path, monkeypatch):
    exe = tmp_path / 'tmc3'
    exe.write_text('#!/bin/sh\necho boom >&2\nexit 7\n')
    exe.chmod(0o755)
    monkeypatch.setenv('PCGC_TMC3', str(exe))
    from pcgcv2_amd.coder import CoordinateCoder
    with pytest.raises(RuntimeError, match='tmc3 encode failed'):
        CoordinateCoder(str(tmp_path / 'x')).encode(np.zeros((3, 3), np.int32))
    assert not [p for p in tmp_path.iterdir() if p.suffix == '.ply']


def test_numa_cpu_lookup_and_thread_budget(tmp_path):
    """pcgcv2_amd.numa_cpus_of_gpu reads the GPU's NUMA node and that node's CPU list from sysfs (faked here);
    configure_host_threads budgets frames x (launcher + helper + range-decoder threads + ATen threads) within the CPU quota."""
    import pcgcv2_amd
    assert pcgcv2_amd._parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    dev = tmp_path / 'bus/pci/devices/0000:c1:00.0'
    dev.mkdir(parents=True)
    (dev / 'numa_node').write_text('1\n')
    node = tmp_path / 'devices/system/node/node1'
    node.mkdir(parents=True)
    (node / 'cpulist').write_text('16-31,144-159\n')
    cpus = pcgcv2_amd.numa_cpus_of_gpu('0000:C1:00.0', sysfs=str(tmp_path))
    assert cpus == list(range(16, 32)) + list(range(144, 160))
    (dev / 'numa_node').write_text('-1\n')
    assert pcgcv2_amd.numa_cpus_of_gpu('0000:c1:00.0', sysfs=str(tmp_path)) is None
    assert pcgcv2_amd.numa_cpus_of_gpu('0000:ff:00.0', sysfs=str(tmp_path)) is None
    try:
        quota = pcgcv2_amd.effective_cpus()
        for frames in (1, 2, 4, 8):
            cfg = pcgcv2_amd.configure_host_threads(frames_in_flight=frames)
            assert cfg['rc_threads'] >= 1 and cfg['aten_threads'] >= 1
            if frames > 1 and quota >= 4 * frames:
                assert frames * (2 + (cfg['rc_threads'] - 1) + cfg['aten_threads']) <= quota + frames     # (the decoding thread is one of rc_threads)
    finally:
        pcgcv2_amd.configure_host_threads()


def test_reference_autoencoder_binds_to_the_me_facade(golden_dir):
    """`pcgcv2_amd.ME` stands in for MinkowskiEngine: the REFERENCE's autoencoder.py, loaded as it is (this container only; the GPU
    box has no /root/reference), must construct its Encoder / Decoder from the facade's layers, and their parameters must be exactly
    the product's checkpoint keys and shapes — i.e. a reference checkpoint loads into the reference's own module tree built on these
    operators."""
    import importlib.util, sys, types
    ref = '/root/reference/autoencoder.py'
    if not os.path.exists(ref):
        pytest.skip('reference sources are not available here')
    import pcgcv2_amd.ME as ME
    import pcgcv2_amd.data_utils as du
    saved = {k: sys.modules.get(k) for k in ('MinkowskiEngine', 'data_utils')}
    sys.modules['MinkowskiEngine'], sys.modules['data_utils'] = ME, du
    try:
        spec = importlib.util.spec_from_file_location('_ref_autoencoder', ref)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        enc, dec = mod.Encoder(channels=[1, 16, 32, 64, 32, 8]), mod.Decoder(channels=[8, 64, 32, 16])
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    from pcgcv2_amd.pcc_model import PCCModel
    got = {('encoder.' + k): tuple(v.shape) for k, v in enc.state_dict().items()}
    got.update({('decoder.' + k): tuple(v.shape) for k, v in dec.state_dict().items()})
    want = {k: tuple(v.shape) for k, v in PCCModel().state_dict().items() if k.split('.')[0] in ('encoder', 'decoder')}
    assert got == want and len(got) == 212                        # the 227 checkpoint keys minus the 15 entropy-bottleneck ones (golden G5)
    assert got['encoder.conv0.kernel'] == (27, 1, 16) and got['encoder.down0.kernel'] == (8, 16, 32) and got['encoder.block0.0.conv1_0.kernel'] == (32, 8)


def test_native_item_files_equal_the_python_path(tmp_path):
    """pcgc_items_encode / _probe / _decode (the host half of coding a batch, on native threads) against the Python path item by item:
    every file byte-identical (`_F.bin`, `_H.bin`, `_num_points.bin`, `_C.bin`, the `_F.idx` sidecar with and without checkpoints),
    the decoded symbols and coordinates equal, a foreign table CRC refused, INDEX_SEGMENTS = 0 honoured."""
    import torch
    from pcgcv2_amd import coder
    from pcgcv2_amd.entropy_model import EntropyBottleneck
    rng = np.random.default_rng(21)
    torch.manual_seed(3)
    eb = EntropyBottleneck(8)
    with torch.no_grad():
        for f in eb._factors:
            f.copy_(torch.rand(f.shape) - 0.5)
    rows = [3000, 40, 20000, 1, 700]                      # 20000 rows: a stream with checkpoints; 1 row: the smallest item
    ranges = [(-7.0, 9.0), (0.0, 0.0), (-20.0, 22.0), (-1.0, 1.0), (-3.0, 2.0)]
    syms, xyzs = [], []
    for r, (lo, hi) in zip(rows, ranges):
        L = int(hi - lo) + 1
        s_ = np.clip(np.rint(rng.normal(L / 2, L / 6, size=(r, 8))), 0, L - 1).astype(np.int16)
        s_[0, 0], s_[-1, -1] = 0, L - 1
        syms.append(s_)
        extent = 600 if r == 700 else 60                       # (one item with 10-bit coordinates: the sorted-level builder's general path)
        xyzs.append(rng.permutation(np.unique(rng.integers(0, extent, size=(4 * r + 8, 3)), axis=0))[:r].astype(np.int32))
    rows = [len(x) for x in xyzs]
    syms = [s_[:r] for s_, r in zip(syms, rows)]
    counts = [(r * 3, r * 9, r * 30) for r in rows]
    py, nat = tmp_path / 'py', tmp_path / 'nat'
    py.mkdir(); nat.mkdir()
    default_segments = coder.INDEX_SEGMENTS
    for seg in (default_segments, 8, 3, 0):                  # (3: an odd number of segments — the two-chain decoder's last task is a single)
        coder.INDEX_SEGMENTS = seg
        try:
            fc, cc = coder.FeatureCoder(str(py / 'c'), eb), coder.CoordinateCoder(str(py / 'c'))
            for i, r in enumerate(rows):
                fc.encode_symbols(syms[i], np.float32(ranges[i][0]), np.float32(ranges[i][1]), postfix=f'_{i}')
                cc.encode(xyzs[i], postfix=f'_{i}')
                coder._dump(str(py / f'c_{i}_num_points.bin'), coder._COUNTS.pack(*counts[i]))
            stems = [str(nat / f'c_{i}') for i in range(len(rows))]
            ops.items_encode(stems, np.concatenate(syms), np.concatenate(xyzs), rows, ranges, counts, eb._host_packed(), seg)
            for i in range(len(rows)):
                for suffix in coder.STREAMS + ('_F.idx',):
                    a, b = py / f'c_{i}{suffix}', nat / f'c_{i}{suffix}'
                    assert a.exists() == b.exists(), (seg, i, suffix)
                    if a.exists():
                        assert a.read_bytes() == b.read_bytes(), (seg, i, suffix)
            got_rows, C, got_ranges, got_counts, native = ops.items_probe(stems)
            assert list(got_rows) == rows and C == 8 and native.all()
            np.testing.assert_array_equal(got_ranges, np.asarray(ranges, np.float32))
            np.testing.assert_array_equal(got_counts, np.asarray(counts, np.int32))
            sym, xyz = ops.items_decode(stems, got_rows, C, got_ranges, native, eb._host_packed(), use_sidecar=bool(seg))
            np.testing.assert_array_equal(sym, np.concatenate(syms))
            off = 0
            for i, r in enumerate(rows):                  # the octree codec returns the voxels in its own order
                assert set(map(tuple, xyz[off:off + r])) == set(map(tuple, xyzs[i]))
                off += r
            # coord_layout 1: the coordinate LEVEL Coder.decode starts from (coder.py:97-102) — rows (item, 8x, 8y, 8z), every item in
            # sort_spare_tensor's (z, y, x) order (data_utils.py:91-101), written into a caller's buffer
            buf = np.full((sum(rows) + 5, 4), -7, np.int32)
            sym2, level = ops.items_decode(stems, got_rows, C, got_ranges, native, eb._host_packed(), use_sidecar=bool(seg), level_scale=8, level_out=buf)
            np.testing.assert_array_equal(sym2, sym)
            assert level.base is buf or level is buf or np.shares_memory(level, buf)
            assert (buf[sum(rows):] == -7).all()
            off = 0
            for i, r in enumerate(rows):
                want = xyzs[i][np.lexsort((xyzs[i][:, 0], xyzs[i][:, 1], xyzs[i][:, 2]))]
                np.testing.assert_array_equal(level[off:off + r, 0], i)
                np.testing.assert_array_equal(level[off:off + r, 1:], want * 8)
                off += r
            with pytest.raises(PcgcError):
                ops.items_decode(stems, got_rows, C, got_ranges, native, eb._host_packed(), level_scale=8, level_out=np.zeros((3, 4), np.int32))
            # pcgc_frame_decode: one cloud, probe + both streams in one call into the caller's buffers; too small -> the size needed
            off = 0
            for i, r in enumerate(rows):
                sym_buf, level_buf = np.full((r + 3, 8), -9, np.int16), np.full((r + 3, 4), -9, np.int32)
                n, rng_i, counts_i, nat_i = ops.frame_decode(stems[i], 8, eb._host_packed(), sym_buf, level_buf, use_sidecar=bool(seg))
                assert (n, nat_i, counts_i) == (r, True, tuple(counts[i])) and tuple(float(v) for v in rng_i) == tuple(ranges[i])
                np.testing.assert_array_equal(sym_buf[:r], syms[i])
                np.testing.assert_array_equal(level_buf[:r, 1:], level[off:off + r, 1:])
                assert (level_buf[:r, 0] == 0).all() and (sym_buf[r:] == -9).all() and (level_buf[r:] == -9).all()
                if r > 1:
                    assert ops.frame_decode(stems[i], 8, eb._host_packed(), sym_buf[:r - 1], level_buf[:r - 1]) == (r, None, None, None)
                off += r
            with pytest.raises(PcgcError, match='channels'):
                ops.frame_decode(stems[0], 4, eb._host_packed(), np.zeros((rows[0], 4), np.int16), np.zeros((rows[0], 4), np.int32))
            with pytest.raises(PcgcError, match='_H.bin'):           # no such cloud
                ops.frame_decode(str(nat / 'nothing_here'), 8, eb._host_packed(), np.zeros((4, 8), np.int16), np.zeros((4, 4), np.int32))
        finally:
            coder.INDEX_SEGMENTS = default_segments
    # a sidecar that names another table: refused (the 20000-row item of the first pass is gone; re-encode one item)
    ops.items_encode([str(nat / 'g')], syms[0], xyzs[0], rows[:1], ranges[:1], counts[:1], eb._host_packed(), 8)
    blob = bytearray((nat / 'g_F.idx').read_bytes())
    import struct, zlib
    blob[16] ^= 0xFF
    blob[20:24] = struct.pack('<I', zlib.crc32(bytes(blob[:20]) + bytes(blob[24:])))
    (nat / 'g_F.idx').write_bytes(bytes(blob))
    r1, C1, rg1, ct1, nv1 = ops.items_probe([str(nat / 'g')])
    with pytest.raises(PcgcError, match='CDF table'):
        ops.items_decode([str(nat / 'g')], r1, C1, rg1, nv1, eb._host_packed())
    ops.items_decode([str(nat / 'g')], r1, C1, rg1, nv1, eb._host_packed(), use_sidecar=False)      # without the sidecar: decodes


def test_items_decode_refuses_a_foreign_channel_count(tmp_path):
    """a damaged / foreign `_H.bin` names C = 4096 channels; the library sizes the table and the parameter read (44 * C floats)
    from whatever C it is handed, so a header's C must never travel past the binding with the model's 44 * 8 parameters."""
    import struct
    from pcgcv2_amd.entropy_model import EntropyBottleneck
    rng = np.random.default_rng(5)
    torch.manual_seed(5)
    eb = EntropyBottleneck(8)
    r = 500
    sym = np.clip(np.rint(rng.normal(5, 2.0, size=(r, 8))), 0, 10).astype(np.int16)
    sym[0, 0], sym[-1, -1] = 0, 10
    xyz = rng.permutation(np.unique(rng.integers(0, 40, size=(4 * r, 3)), axis=0))[:r].astype(np.int32)
    stem = str(tmp_path / 'h')
    ops.items_encode([stem], sym, xyz, [r], [(-5.0, 5.0)], [(r, 2 * r, 3 * r)], eb._host_packed(), 16)
    head = bytearray(open(stem + '_H.bin', 'rb').read())
    head[4:8] = struct.pack('<i', 4096)
    open(stem + '_H.bin', 'wb').write(bytes(head))
    rows, C, ranges, counts, native = ops.items_probe([stem])
    assert C == 4096
    with pytest.raises(PcgcError, match='entropy parameters'):
        ops.items_decode([stem], rows, C, ranges, native, eb._host_packed())
    with pytest.raises(PcgcError, match='entropy parameters'):
        ops.items_encode([stem], sym[:, :4].copy(), xyz, [r], [(-5.0, 5.0)], [(r, 2 * r, 3 * r)], eb._host_packed(), 16)


def test_table_cache_policy(tmp_path):
    """pcgc_table_cache / entropy_model.table_cache: dropping or disabling the caches changes no byte — only who evaluates the table."""
    from pcgcv2_amd import entropy_model
    from pcgcv2_amd.entropy_model import EntropyBottleneck
    from pcgcv2_amd._lib import lib
    rng = np.random.default_rng(8)
    torch.manual_seed(8)
    eb = EntropyBottleneck(8)
    r = 2500
    sym = np.clip(np.rint(rng.normal(6, 2.0, size=(r, 8))), 0, 12).astype(np.int16)
    sym[0, 0], sym[-1, -1] = 0, 12
    xyz = rng.permutation(np.unique(rng.integers(0, 50, size=(4 * r, 3)), axis=0))[:r].astype(np.int32)

    def code(stem):
        ops.items_encode([stem], sym, xyz, [r], [(-6.0, 6.0)], [(r, 2 * r, 3 * r)], eb._host_packed(), 16)
        sb, lb = np.zeros((r, 8), np.int16), np.zeros((r, 4), np.int32)
        ops.frame_decode(stem, 8, eb._host_packed(), sb, lb)
        np.testing.assert_array_equal(sb, sym)
        return {k: open(stem + k, 'rb').read() for k in ('_F.bin', '_H.bin', '_F.idx')}
    try:
        entropy_model.table_cache(on=True, clear=True)
        a = code(str(tmp_path / 'a'))
        assert lib().pcgc_table_cache(0) >= 1                       # the table of (parameters, -6..6) was cached ... and is dropped now
        assert lib().pcgc_table_cache(0) == 0
        b = code(str(tmp_path / 'b'))                               # evaluated afresh: same bytes
        entropy_model.table_cache(on=False)
        c = code(str(tmp_path / 'c'))                               # nothing is kept ...
        assert lib().pcgc_table_cache(0) == 0
        t1, crc1 = eb.host_table(np.float32(-6), np.float32(6), None, want_crc=True)
        assert '_table_cache' not in eb.__dict__ or not eb.__dict__['_table_cache']
        entropy_model.table_cache(on=True)
        t2, crc2 = eb.host_table(np.float32(-6), np.float32(6), None, want_crc=True)
        assert eb.host_table(np.float32(-6), np.float32(6), None) is t2 and crc1 == crc2 and np.array_equal(t1, t2)
        assert entropy_model.table_cache(clear=True) >= 0 and not eb.__dict__.get('_table_cache')
        assert a == b == c
    finally:
        entropy_model.table_cache(on=True)


def test_host_table_cache_under_concurrent_misses():
    """compress_symbols / decompress_symbols run on pool threads; with a full cache, concurrent misses used to evict the same
    oldest key (KeyError) or trip over a dict that changed size.  32 threads x 40 distinct ranges against a 16-entry cache."""
    from concurrent.futures import ThreadPoolExecutor
    from pcgcv2_amd.entropy_model import EntropyBottleneck
    torch.manual_seed(2)
    eb = EntropyBottleneck(8)

    def work(t):
        for i in range(40):
            lo = -float(1 + (i * 7 + t) % 23)
            tab = eb.host_table(np.float32(lo), np.float32(5.0), None)
            assert tab.shape == (8, int(5.0 - lo) + 2)
        return True
    with ThreadPoolExecutor(16) as ex:
        assert all(ex.map(work, range(32)))
    assert len(eb.__dict__['_table_cache']) <= eb.TABLE_CACHE_SIZE


@pytest.mark.parametrize('name,n', [('solid_cube_s', 13824), ('solid_ball_s', 14592), ('noisy_s', 11086), ('multi_s', 15733), ('sparse_s', 7571)])
def test_geometry_families_on_the_oracle(name, n):
    """The non-shell synthetic clouds (synthetic.CLOUDS: filled bodies with masses of exactly tied logits, noisy / holed surfaces,
    several components with one-voxel sheets and rods, a thinned surface): deterministic, unique, and the oracle codes them — the
    bitstream does not depend on the row order of the input (the latent is sorted, coder.py:83), the decoded COUNT follows the budgets
    in either order, and up-sampling (rho = 4, coder.py:107) keeps min(4 N1, candidates) voxels."""
    p = synthetic.cloud(name).numpy()
    assert len(p) == n and len(np.unique(p, axis=0)) == n and p.min() >= 0
    q = synthetic.cloud(name, order='shuffled', seed=3).numpy()
    assert not np.array_equal(p, q) and set(map(tuple, p)) == set(map(tuple, q))
    np.testing.assert_array_equal(q, synthetic.cloud(name, order='shuffled', seed=3).numpy())
    sd = synthetic.state_dict_to_numpy(synthetic.synthetic_state_dict())
    encs = []
    for pts in (p, q):
        c4 = np.concatenate([np.zeros((len(pts), 1), np.int32), pts], 1)
        enc = orc.encode(sd, c4)
        out = orc.decode(sd, enc['coords8'], enc['F'], enc['H'], enc['num_points'])
        assert len(out) == n and len(np.unique(out, axis=0)) == n
        encs.append(enc)
    for k in ('F', 'H', 'num_points'):
        assert encs[0][k] == encs[1][k], k
    n4, n2, n1 = np.frombuffer(encs[0]['num_points'], np.int32)
    up = orc.decode(sd, encs[0]['coords8'], encs[0]['F'], encs[0]['H'], encs[0]['num_points'], rho=4.0)
    assert len(up) == min(4 * n1, 8 * n2) and len(np.unique(up, axis=0)) == len(up)


def test_full_size_geometry_families_have_the_documented_sizes():
    for name, n in (('solid_cube', 512000), ('solid_ball', 539152), ('multi10', 1022977)):
        assert len(synthetic.cloud(name)) == n, name


@pytest.mark.parametrize('threads', [1, 2, 8])
def test_lane_parallel_range_decoder_equals_the_serial_one(threads):
    """pcgc_set_rc_lanes(1): eight segments of an indexed stream per zmm register on the calling thread (the decoder of ranks with a one- or
    two-thread budget).  Alphabets of 1 ... 63 symbols, 8 / 12 / 16 checkpoints, ragged row counts, peaked / uniform / minimum-probability
    tables: the symbols equal those of the serial decoder and the bit-serial oracle; with fewer than eight checkpoints or an alphabet
    beyond the SIMD limit the call falls back by itself."""
    rng = np.random.default_rng(42)
    cases = []
    for L, n, ck in ((21, 20000, 16), (21, 20003, 16), (2, 9000, 16), (63, 12000, 16), (5, 8001, 8), (21, 15000, 12), (40, 9999, 16), (21, 4000, 16),
                     (100, 9000, 16), (21, 5000, 4), (1, 3000, 16)):
        pmf = rng.random((8, L)) ** 3 + 1e-4
        if L >= 5:
            pmf[:, L // 2] += 3.0                                             # a peak, like real latents
            pmf[:, 0] = 1e-7                                                  # and a minimum-probability symbol (17-bit shifts)
        table = _table_from_pmf(pmf)
        p = pmf / pmf.sum(1, keepdims=True)
        sym = np.stack([rng.choice(L, size=n, p=p[c]) for c in range(8)], 1).astype(np.int16)
        sym[0, 0], sym[-1, -1] = 0, L - 1
        cases.append((table, sym, ck))
    ops.set_rc_threads(threads)
    try:
        for table, sym, ck in cases:
            data, index = ops.rc_encode(table, sym, checkpoints=ck)
            assert data == orc.rc_encode(table, sym)
            ops.set_rc_lanes(0)
            ref = ops.rc_decode(table, data, sym.size, index=index)
            ops.set_rc_lanes(1)
            got = ops.rc_decode(table, data, sym.size, index=index)
            np.testing.assert_array_equal(ref, sym.ravel())
            np.testing.assert_array_equal(got, sym.ravel())
            ops.set_rc_lanes(-1)
            np.testing.assert_array_equal(ops.rc_decode(table, data, sym.size, index=index), sym.ravel())
    finally:
        ops.set_rc_lanes(-1)
        ops.set_rc_threads(0)


def test_frame_decode_in_two_halves(tmp_path):
    """pcgc_frame_decode_begin / _end: the same symbols and level as the one-call form; the coordinate level is complete when `_begin`
    returns; a damaged feature stream surfaces in `_end`; too-small buffers leave nothing pending; four threads at once (one is served
    asynchronously, the others synchronously) all get their own clouds."""
    import threading
    from pcgcv2_amd.entropy_model import EntropyBottleneck
    torch.manual_seed(4)
    eb = EntropyBottleneck(8)
    packed = eb._host_packed()
    clouds = []
    for t in range(4):
        rng = np.random.default_rng(200 + t)
        r = 5000 + 2500 * t
        sym = np.clip(np.rint(rng.normal(7, 2.0, size=(r, 8))), 0, 14).astype(np.int16)
        sym[0, 0], sym[-1, -1] = 0, 14
        xyz = rng.permutation(np.unique(rng.integers(0, 80 + 10 * t, size=(4 * r, 3)), axis=0))[:r].astype(np.int32)
        stem = str(tmp_path / f'c{t}')
        ops.items_encode([stem], sym, xyz, [r], [(-7.0, 7.0)], [(r, 2 * r, 3 * r)], packed, 16)
        clouds.append((stem, r, sym, xyz[np.lexsort((xyz[:, 0], xyz[:, 1], xyz[:, 2]))] * 8))
    stem, r, sym, want = clouds[1]
    for _ in range(5):
        sb, lb = np.full((r + 2, 8), -5, np.int16), np.full((r + 2, 4), -5, np.int32)
        n, rng_, counts, native = ops.frame_decode_begin(stem, 8, packed, sb, lb)
        assert (n, native, counts) == (r, True, (r, 2 * r, 3 * r)) and tuple(float(v) for v in rng_) == (-7.0, 7.0)
        np.testing.assert_array_equal(lb[:r, 1:], want)          # the level is complete HERE; the symbols only after _end
        ops.frame_decode_end()
        np.testing.assert_array_equal(sb[:r], sym)
        assert (sb[r:] == -5).all() and (lb[r:] == -5).all()
    ops.frame_decode_end()                                       # nothing pending: a no-op
    assert ops.frame_decode_begin(stem, 8, packed, sb[:r - 1], lb[:r - 1]) == (r, None, None, None)
    ops.frame_decode_end()
    # a damaged feature stream: the error arrives in _end (or already in _begin when the call was served synchronously)
    blob = open(stem + '_F.bin', 'rb').read()
    os.remove(stem + '_F.bin')                                   # a missing feature stream: only the feature task can notice
    with pytest.raises(PcgcError, match='_F.bin'):
        ops.frame_decode_begin(stem, 8, packed, sb, lb)
        ops.frame_decode_end()
    ops.frame_decode_end()
    open(stem + '_F.bin', 'wb').write(blob)
    errors = []

    def worker(t):
        stem_t, r_t, sym_t, want_t = clouds[t]
        sbt, lbt = np.zeros((r_t, 8), np.int16), np.zeros((r_t, 4), np.int32)
        for _ in range(10):
            sbt[:] = -1; lbt[:] = -1
            n_t, _, _, nat = ops.frame_decode_begin(stem_t, 8, packed, sbt, lbt, use_sidecar=(t != 1))
            ok_level = np.array_equal(lbt[:, 1:], want_t)
            ops.frame_decode_end()
            if n_t != r_t or not nat or not ok_level or not np.array_equal(sbt, sym_t):
                errors.append(t)
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    # a worker that is late to its job (a busy host): `_end` runs the feature stream on the calling thread; the late worker finds nothing
    import time
    before = lib().pcgc_frame_worker_test(20000)
    try:
        for _ in range(3):
            sb, lb = np.full((r, 8), -5, np.int16), np.full((r, 4), -5, np.int32)
            ops.frame_decode_begin(stem, 8, packed, sb, lb)
            np.testing.assert_array_equal(lb[:, 1:], want)
            ops.frame_decode_end()
            np.testing.assert_array_equal(sb, sym)
        assert lib().pcgc_frame_worker_test(-1) >= before + 3
    finally:
        lib().pcgc_frame_worker_test(0)
    time.sleep(0.05)                                             # (the late worker wakes up to an empty slot)
    sb, lb = np.full((r, 8), -5, np.int16), np.full((r, 4), -5, np.int32)
    ops.frame_decode_begin(stem, 8, packed, sb, lb)
    ops.frame_decode_end()
    np.testing.assert_array_equal(sb, sym)


def test_dispatch_table_rules_without_a_gpu():
    """the policy table (pcgcv2_amd/dispatch.py) is host logic: the families of the round-4 kernels, their gates and their fall-backs
    (32-bit offset limit, residual / out= forms the packed kernel does not have) without a device"""
    from pcgcv2_amd import dispatch, ops
    assert dispatch.select('prune', (16,), 2_000_000).family == 'select'
    assert dispatch.select('prune', (1,), 1000).family == 'mask'                       # (features that are not whole 16-byte chunks)
    assert dispatch.select('conv3', (64, 64), ops.PACKED_CONV64_MIN).family == 'packed'
    assert dispatch.select('conv3', (64, 64), ops.PACKED_CONV64_MIN - 1).family == 'gather'
    assert dispatch.select('conv3', (64, 64), 100_000, plain_output=False).family == 'gather'
    assert dispatch.select('conv3', (64, 64), 100_000, extent=dispatch.LIMIT).family == 'gather'
    assert dispatch.select('conv3', (64, 64), 100_000, 'children').family == 'packed'    # (the decoder's conv0: a children level with its own map)
    assert dispatch.select('conv3', (32, 32), 100_000).family == 'rows'
    assert dispatch.select('conv3', (1, 16), 100_000, unit_input=True).family == 'unit'
    keep = {k: getattr(ops, k) for k in ('ONE_SWEEP_PRUNE', 'PACKED_CONV64')}
    try:
        ops.ONE_SWEEP_PRUNE = ops.PACKED_CONV64 = False
        assert dispatch.select('prune', (16,), 2_000_000).family == 'mask'
        assert dispatch.select('conv3', (64, 64), 100_000).family == 'gather'
    finally:
        for k, v in keep.items():
            setattr(ops, k, v)
    assert len(dispatch.describe().splitlines()) == len(dispatch.TABLE)


def test_q4x_schedules_counted_waits(tmp_path):
    """The quad-block engine (csrc/q4x.h) waits for its LDS-DMA gathers with COUNTED `s_waitcnt vmcnt(N)` taken from a compile-time simulation
    of the VMEM issue order (csrc/q4x_sched.h): N too large is a data race on the gather ring, N too small a stall.  The schedules of every
    instantiation the library launches are dumped by a g++ build of the same headers and checked against an independent re-simulation of the
    engine's issue order: every cell gathered exactly once, into a ring slot whose previous rows are already in registers; every wait
    exactly the number of gather instructions issued after the awaited cell's; counts within vmcnt's 6 bits; per accumulator the weight
    fragments in the order of the canonical chain (ascending offset, then ascending channel half)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / 'q4x_dump')
    subprocess.run(['g++', '-std=c++17', '-O1', '-o', exe, os.path.join(root, 'tests', 'native', 'q4x_sched_dump.cpp')], check=True)
    lines = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines()
    assert len(lines) == 6
    for line in lines:
        S = json.loads(line)
        MT, D, nc, groups = S['MT'], S['D'], S['ncells'], S['groups']
        per = 4 * MT                                              # gather instructions per cell
        # structure: cells in order, first / last flags
        cells_of = [g['cell'] for g in groups]
        assert cells_of == sorted(cells_of) and set(cells_of) == set(range(nc))
        for i, g in enumerate(groups):
            assert g['first'] == (i == 0 or groups[i - 1]['cell'] != g['cell'])
            assert g['last'] == (i == len(groups) - 1 or groups[i + 1]['cell'] != g['cell'])
        first_group = {g['cell']: i for i, g in reversed(list(enumerate(groups)))}
        last_group = {g['cell']: i for i, g in enumerate(groups)}
        # the engine's order: prologue gathers cells 0 .. D-1; per group: (first: the cell's planned gathers) then (last: wait for cell + 1, read it)
        issued_at, order, ops = {}, [], 0                         # cell -> (group index or -1, ops count after its last instruction)
        for c in range(min(D, nc)):
            ops += per; issued_at[c] = (-1, ops); order.append(c)
        rows_in_regs = {0: -1}                                    # cell -> group index at whose START its rows are requested into registers (-1: prologue)
        for i, g in enumerate(groups):
            c = g['cell']
            if g['first']:
                for tgt in S['cells'][c][3:5]:
                    if tgt >= 0:
                        assert tgt not in issued_at, 'a cell gathered twice'
                        # its ring slot held cell tgt - D: those rows must be in registers — requested in an EARLIER group and waited for (lgkmcnt(0)) at this group's start
                        assert tgt - D in rows_in_regs and rows_in_regs[tgt - D] < i
                        ops += per; issued_at[tgt] = (i, ops); order.append(tgt)
            if g['last'] and c + 1 < nc:
                assert c + 1 in issued_at, 'waiting for a cell that was never gathered'
                want = ops - issued_at[c + 1][1]                  # instructions issued after the awaited cell's last one
                assert g['vm_wait'] == want, (S['name'], MT, D, i, g['vm_wait'], want)
                assert 0 <= want < 64
                rows_in_regs[c + 1] = i
        assert sorted(issued_at) == list(range(nc))
        # chain order per accumulator: (offset, half) ascending; slots of one group ascending in (row quarter) per accumulator
        seen = {}
        for g in groups:
            kp, _, byte_off = S['cells'][g['cell']][:3]
            for e in range(4):
                if g['acc'][e] < 0:
                    continue
                key = (kp, byte_off, g['rowq'][e])
                assert seen.get(g['acc'][e], (-1, -1, -1)) < key, (S['name'], g)
                seen[g['acc'][e]] = key
        if S['name'] == 'A32':
            assert len(groups) == 112 and nc == 54 and sorted(seen) == [0, 1, 2, 3]
            assert [g['frag'] for g in groups if g['acc'][0] == 0] == [(k * 2 + h) * 2 for k in range(27) for h in range(2)]
            assert [g['frag'] for g in groups if g['acc'][0] == 3] == [(27 * 2 + h) * 2 + 1 for h in range(2)]
        else:
            assert len(groups) == 81 and nc == 27 and sorted(seen) == [0, 1, 2, 3, 4, 5]
            assert all(g['frag'] == 3 * g['cell'] + j for g, j in zip(groups, [0, 1, 2] * 27))

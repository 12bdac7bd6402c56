"""Codec API of PCGCv2 on the MI355X operator set — drop-in for the reference's `coder.py` (classes CoordinateCoder,
FeatureCoder, Coder at coder.py:16-112; CLI at coder.py:114-184): same constructor / method signatures, same four files.

Bitstream of one coded cloud, `<prefix><postfix>` + :
    _C.bin            coordinates of the stride-8 latent: a G-PCC stream if a `tmc3` binary is installed, else the native
                      "PCGO" octree stream (gpcc.py)
    _F.bin            range-coded latent features (torchac-compatible stream)
    _H.bin            17 bytes, little endian:  int32 N8 | int32 C | int8 1 | float32 min_v | float32 max_v
    _num_points.bin   int32[3] = [N4, N2, N1]   (the top-k budgets of the three decoder stages)
"""
import os
import struct
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import gpcc, ops
from .data_utils import (array2vector, istopk, sort_spare_tensor, load_sparse_tensor, scale_sparse_tensor,  # noqa: F401
                         write_ply_ascii_geo, read_ply_ascii_geo)
from .pc_error import pc_error
from .pcc_model import PCCModel
from .sparse import SparseTensor, CoordMap, require_gpu

def _os_environ_flag(name, default='1'):
    return os.environ.get(name, default) != '0'


device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
_POOL = ThreadPoolExecutor(max_workers=4, thread_name_prefix='pcgc-coord')      # helpers of up to 4 frames in flight

_HEADER = struct.Struct('<iibff')                 # (N8, C, len(min_v)=1, min_v, max_v) — coder.py:51-55
_COUNTS = struct.Struct('<3i')                    # (N4, N2, N1)                       — coder.py:85-87
STREAMS = ('_C.bin', '_F.bin', '_H.bin', '_num_points.bin')


def _dump(path, payload):
    with open(path, 'wb') as fh:
        fh.write(payload)


def _slurp(path):
    with open(path, 'rb') as fh:
        return fh.read()


# ---- sidecar of `_F.bin` (not part of the reference's format; see FeatureCoder) ---------------------------------------------------
# (1) Decoding index.  A range-coded stream is sequential only because the decoder state at a later symbol is unknown.  The
# encoder knows it: next to `_F.bin` (bit-identical to the reference's stream, decodable by torchac and by this package without
# the sidecar) it writes `_F.idx` — the decoder state at INDEX_SEGMENTS row boundaries, 24 bytes each — and the decoder decodes
# the segments on several threads (1.4 ms -> ~0.3 ms for the 150 k latent symbols of a vox10 frame).
# (2) Table guard.  The uint16 CDF table is evaluated with torch's CPU kernels (the reference's arithmetic, entropy_model.py
# docstring), which are not bit-reproducible across CPU kinds / torch builds: a stream coded on one host can meet a table that
# differs by one count on another and decode to garbage without any error.  The sidecar carries the CRC-32 of the encoder's
# table; a decoder that derives a different table REFUSES the stream instead of returning noise.
# The sidecar names the stream it belongs to by length and CRC-32 and carries a CRC over itself; one that does not match (or is
# absent: a reference-made stream) is ignored and the stream is decoded serially, unguarded — exactly what the reference does.
INDEX_SEGMENTS = 16                              # checkpoints per stream (two segments per decoder thread); 0 = never write or read the sidecar
FRAME_SPLIT = _os_environ_flag('PCGC_FRAME_SPLIT')   # decode: coordinate-only kernels are enqueued while the feature stream is still being decoded (A/B: PCGC_FRAME_SPLIT=0)
INGEST_SORT = True                               # encode: an unordered cloud is sorted once before the encoder touches it (Coder._ingest)
TIMELINE = None                                  # measurement hook (bench.py): a list -> (mark, perf_counter) of the points between which the GPU has nothing queued
WARM_TABLE_CODE = True                           # encode: a throw-away table evaluation while the host waits for the GPU (ops.table_warm)
NATIVE_ITEMS = True                              # batches: per-item host stages on native threads (False: Python thread pool; A/B and tests)
INDEX_SUFFIX = '_F.idx'
_INDEX_HEAD = struct.Struct('<4sIIIII')          # magic, stream bytes, stream CRC-32, checkpoints, table CRC-32, CRC-32 of (head so far + body)


def _pack_index(payload, index, table_crc=0):
    import zlib
    idx = np.ascontiguousarray(np.zeros((0, ops.RC_CKPT_WORDS)) if index is None else index, dtype='<u4')
    head = struct.pack('<4sIIII', b'PCG2', len(payload), zlib.crc32(payload), idx.shape[0], int(table_crc) & 0xFFFFFFFF)
    body = idx.tobytes()
    return head + struct.pack('<I', zlib.crc32(head + body)) + body


def _load_sidecar(path, payload):
    """-> (index uint32 [k, RC_CKPT_WORDS] or None, table CRC-32 or None); (None, None) unless the file is intact AND belongs to
    `payload`."""
    import zlib
    try:
        blob = _slurp(path)
    except OSError:
        return None, None
    if len(blob) < _INDEX_HEAD.size:
        return None, None
    magic, nbytes, crc, count, table_crc, self_crc = _INDEX_HEAD.unpack(blob[:_INDEX_HEAD.size])
    if (magic != b'PCG2' or nbytes != len(payload) or len(blob) != _INDEX_HEAD.size + count * 4 * ops.RC_CKPT_WORDS
            or self_crc != zlib.crc32(blob[:_INDEX_HEAD.size - 4] + blob[_INDEX_HEAD.size:]) or crc != zlib.crc32(payload)):
        return None, None
    index = np.frombuffer(blob, dtype='<u4', offset=_INDEX_HEAD.size).reshape(count, ops.RC_CKPT_WORDS) if count >= 2 else None
    return index, table_crc


def _load_index(path, payload):
    return _load_sidecar(path, payload)[0]


def index_bits(prefix, postfix=''):
    """bits of the optional decoding index next to `_F.bin` (0 if none was written)."""
    path = prefix + postfix + INDEX_SUFFIX
    return os.path.getsize(path) * 8 if os.path.exists(path) else 0


def stream_bits(prefix, postfix=''):
    """bits of the four files of one coded cloud (coder.py:169-170)."""
    return np.array([os.path.getsize(prefix + postfix + s) * 8 for s in STREAMS])


class CoordinateCoder():
    """Lossless coder of the stride-8 coordinates.  With tmc3 installed: the reference's temp-PLY + subprocess protocol
    (coder.py:23-36); otherwise the in-process octree codec, no temp files."""

    def __init__(self, filename):
        self.filename = filename
        self.ply_filename = filename + '.ply'             # the reference's fixed temp name (coder.py:21); kept as an attribute only

    def _path(self, postfix):
        return self.filename + postfix + '_C.bin'

    def _temp_ply(self, postfix):
        """A temp PLY of this call alone.  The reference reuses `<filename>.ply` for every call, which is only safe because it
        codes sequentially; here frames are coded concurrently (shard.code_units(in_flight>1), one Coder per worker with the
        same prefix), so every call gets its own file next to the bitstream."""
        import tempfile
        head, tail = os.path.split(self.filename + postfix)
        fd, path = tempfile.mkstemp(prefix=tail + '_C_', suffix='.ply', dir=head or '.')
        os.close(fd)
        return path

    def encode(self, coords, postfix=''):
        pts = coords.numpy() if isinstance(coords, torch.Tensor) else np.asarray(coords)
        pts = pts.astype('int')
        if gpcc.tmc3_path() is None:
            gpcc.native_encode(pts, self._path(postfix))
            return
        ply = self._temp_ply(postfix)
        try:
            write_ply_ascii_geo(filedir=ply, coords=pts)
            gpcc.gpcc_encode(ply, self._path(postfix))
        finally:
            os.remove(ply)

    def decode(self, postfix=''):
        if gpcc.is_native_stream(self._path(postfix)):
            return gpcc.native_decode(self._path(postfix))
        ply = self._temp_ply(postfix)
        try:
            gpcc.gpcc_decode(self._path(postfix), ply)
            return read_ply_ascii_geo(ply)
        finally:
            os.remove(ply)


class FeatureCoder():
    """Latent features <-> `_F.bin` + `_H.bin` through the factorized entropy bottleneck (coder.py:39-70)."""

    def __init__(self, filename, entropy_model):
        self.filename = filename
        self.entropy_model = entropy_model.cpu()      # the reference moves it to the CPU; ours stays on the GPU (no-op)

    def encode(self, feats, postfix=''):
        n, c = feats.shape
        segments = min(int(INDEX_SEGMENTS), n // 1024)           # (a segment shorter than ~8 k symbols is not worth a checkpoint)
        info = {}
        if segments >= 2:
            payload, min_v, max_v, index = self.entropy_model.compress(feats, checkpoints=segments, info=info)
        else:
            payload, min_v, max_v = self.entropy_model.compress(feats, info=info)
            index = None
        self._write(postfix, n, c, payload, min_v, max_v, index, info['table_crc'])

    def encode_symbols(self, sym_h, min_v, max_v, postfix='', device=None):
        """encode() from the host side on: int16 symbols [n, c] and their range (one item of a batch, coder.Coder.encode_batch)."""
        n, c = sym_h.shape
        segments = min(int(INDEX_SEGMENTS), n // 1024)
        info = {}
        if segments >= 2:
            payload, min_v, max_v, index = self.entropy_model.compress_symbols(sym_h, min_v, max_v, checkpoints=segments, info=info, device=device)
        else:
            payload, min_v, max_v = self.entropy_model.compress_symbols(sym_h, min_v, max_v, info=info, device=device)
            index = None
        self._write(postfix, n, c, payload, min_v, max_v, index, info['table_crc'])

    def _write(self, postfix, n, c, payload, min_v, max_v, index, table_crc):
        index_path = self.filename + postfix + INDEX_SUFFIX
        if INDEX_SEGMENTS:
            _dump(index_path, _pack_index(payload, index, table_crc))
        elif os.path.exists(index_path):
            os.remove(index_path)                                     # never leave a sidecar of an older stream behind
        _dump(self.filename + postfix + '_F.bin', payload)
        _dump(self.filename + postfix + '_H.bin', _HEADER.pack(n, c, len(min_v), float(min_v[0]), float(max_v[0])))

    def _read(self, postfix):
        n, c, n_minv, min_v, max_v = _HEADER.unpack(_slurp(self.filename + postfix + '_H.bin')[:_HEADER.size])
        if n_minv != 1:
            raise ValueError('unsupported _H.bin: expected one (min_v, max_v) pair')
        payload = _slurp(self.filename + postfix + '_F.bin')
        index, table_crc = _load_sidecar(self.filename + postfix + INDEX_SUFFIX, payload) if INDEX_SEGMENTS else (None, None)
        return n, c, min_v, max_v, payload, index, table_crc

    def decode(self, postfix='', device=None, on_table_launched=None):
        n, c, min_v, max_v, payload, index, table_crc = self._read(postfix)
        return self.entropy_model.decompress(payload, np.float32(min_v), np.float32(max_v), (n, c), channels=c, device=device,
                                             on_table_launched=on_table_launched, index=index, expect_table_crc=table_crc)

    def decode_symbols(self, postfix='', device=None):
        """decode() up to the host symbols -> (int16 ndarray [n, c], min_v) (one item of a batch, coder.Coder.decode_batch)."""
        n, c, min_v, max_v, payload, index, table_crc = self._read(postfix)
        return self.entropy_model.decompress_symbols(payload, np.float32(min_v), np.float32(max_v), (n, c), channels=c, device=device,
                                                     index=index, expect_table_crc=table_crc)


class Coder():
    def __init__(self, model, filename):
        self.model = model
        self.filename = filename
        self.coordinate_coder = CoordinateCoder(filename)
        self.feature_coder = FeatureCoder(self.filename, model.entropy_bottleneck)
        from ._lib import lib
        lib().pcgc_oct_warm()                                   # the coordinate coder's trained priors: built here, not inside the first frame

    @torch.no_grad()
    def encode(self, x, postfix=''):
        """coder.py:80-91: writes the four files, returns the sorted stride-8 latent.  Schedule: the geometry pyramid
        (N1 -> N2 -> N4 -> N8) is built first, so the stride-8 coordinates — all the coordinate coder needs — reach the host
        before the convolutions are even enqueued; the sequential host-side coordinate coding then runs on a helper thread
        while this thread enqueues the encoder and the GPU executes it."""
        with torch.cuda.device(x.device):                       # current device = the tensors' device (kernels, events, streams)
            return self._encode(x, postfix)

    def _ingest(self, x):
        """An UNORDERED cloud (a scanner's or a mesh sampler's PLY: rows in no particular order) is sorted once, into sort_spare_tensor's
        (batch, z, y, x) order, before the encoder touches it.  The canonical row order of every encoder level follows the input order
        (first occurrence), so a random input order drives every gather of every level through random rows (measured on the vox10 frame:
        encode + decode 118 instead of 128 Mpoints/s; the per-row gather kernels alone lose 28 %, profiles/r04_order_probe.txt).  No
        output byte depends on the input order — each output row's fmaf chain follows its neighbours by offset, and the latent is sorted
        before it is coded (coder.py:83) — so the sort changes the speed only.  Ordered-enough input (descents of the key along the
        rows <= n / 64: raster order, already sorted, sorted with a few stragglers) is left alone."""
        d = getattr(x.cmap, 'descents', None)
        if not INGEST_SORT or d is None or d * 64 <= len(x):
            return x
        # one sort per cloud, not per encode: an R-D sweep (test.py) encodes the same tensor once per rate, and the sorted level carries the
        # cached pyramid and kernel maps every rate reuses
        # (kept on the coordinate map, so that CoordMap.drop_caches() — "no geometry survives" — drops it with everything else)
        # The memo holds the feature tensor ITSELF and hits on identity + version: an address-based key would match a NEW tensor the caching
        # allocator placed at a freed one's address (same data_ptr, version 0, same shape) and code the previous cloud's sorted features
        F = x.F
        memo = x.cmap.__dict__.get('_ingested')
        if memo is not None and memo[0] is F and memo[1] == F._version:
            return memo[2]
        order = ops.sort_zyx(x.C, batch_major=True)
        cmap = CoordMap(ops.gather_coords(x.C, order), x.cmap.stride, unique=True)
        cmap.descents = 0
        if x.cmap._batch_rows is not None:
            cmap._batch_rows = list(x.cmap._batch_rows)                  # (batch-major order keeps the items contiguous, in item order)
        if x.has_unit_features():
            y = SparseTensor(x.F, coordinate_map=cmap)                   # all ones: any permutation of it is itself
            y.unit_features, y._unit_stamp = True, x._unit_stamp
        else:
            y = SparseTensor(ops.gather_feats(x.F, order), coordinate_map=cmap)
        x.cmap.__dict__['_ingested'] = (F, F._version, y)
        return y

    def _encode(self, x, postfix):
        x = self._ingest(x)
        lvl8 = x.cmap.build_pyramid(3)                          # cached on the levels: the encoder reuses these maps
        # Side stream: the (z, y, x, batch) order of sort_spare_tensor, the sorted stride-8 coordinates and their copy into pinned
        # memory (N8 x 16 B).  None of it is needed before the latent exists, so the dozen small sort launches run beside the
        # encoder's convolutions instead of in front of them; the helper thread waits for the copy and runs the host coordinate coder.
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(x.device))       # the pyramid is complete after this
        # the encoder's kernel maps, coarse to fine, then its ~40 layer launches (the finest level's own map is only read by the first layer,
        # which on the all-ones input derives presence from the level above: sparse.CoordMap.mapless_unit_conv)
        if x.has_unit_features() and x.cmap.mapless_unit_conv() and ops.PATH.UNIT_INPUT_CONV:
            x.cmap.down()[0].k3
        else:
            x.cmap.k3
        y_list = self.model.encoder(x)                          # the GPU has 2 ms of work queued before the host turns to the side stream
        order, y_C, sorted_ev, arrived, host_C = self._sort_and_stage(lvl8.C, ready)     # (its dozen sort launches run beside the convolutions)
        coded = _POOL.submit(self._encode_geometry, arrived, host_C, lvl8.stride, postfix)     # the octree is coded while the GPU works
        main = torch.cuda.current_stream(x.device)
        main.wait_event(sorted_ev)
        order.record_stream(main)
        y_C.record_stream(main)
        y = SparseTensor(ops.gather_feats(y_list[0].F, order), coordinate_map=CoordMap(y_C, lvl8.stride, unique=True))
        budgets = [len(t) for t in (y_list[1], y_list[2], x)]
        if self._native_items():
            # range + symbols in one synchronising copy; table (cached), range coder, sidecar and the three files in one library call.
            # The host is about to wait for the GPU: it spends that wait pulling the table evaluation's code path into its caches
            # (ops.table_warm: a dummy range, result discarded — the real table is evaluated below, once the range is known)
            if WARM_TABLE_CODE:
                ops.table_warm(self.feature_coder.entropy_model._host_packed(), self.feature_coder.entropy_model._channels)
            min_v, max_v, sym_h = ops.quantize_symbols(y.F)
            if TIMELINE is not None:
                TIMELINE.append(('enc_gpu_done', time.perf_counter()))      # the symbols are on the host: nothing is queued on the GPU from here on
            ops.items_encode([self.filename + postfix], sym_h, np.zeros((0, 3), np.int32), [len(sym_h)], [(min_v, max_v)], [budgets],
                             self.feature_coder.entropy_model._host_packed(), INDEX_SEGMENTS, write_coords=False, threads=1)
        else:
            _dump(self.filename + postfix + '_num_points.bin', _COUNTS.pack(*budgets))
            self.feature_coder.encode(y.F, postfix=postfix)
        coded.result()                                          # (re-raises a coordinate-coder failure)
        return y

    def _sort_and_stage(self, C, ready):
        """On a side stream, once `ready` (C is complete) has fired: canonical order of the rows of C, the sorted coordinates, and
        their asynchronous copy into pinned host memory -> (order, sorted C, event after the sort, event after the copy, pinned
        host view)."""
        dev = C.device
        if getattr(self, '_side', None) is None or self._side.device != dev:
            self._side = torch.cuda.Stream(device=dev)
            self._pinned = None
        if self._pinned is None or self._pinned.numel() < C.numel() or self._pinned.dtype != C.dtype:
            self._pinned = torch.empty(max(C.numel(), 1 << 16), dtype=C.dtype, pin_memory=True)
        host = self._pinned[:C.numel()].view(C.shape)
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            order = ops.sort_zyx(C)
            y_C = ops.gather_coords(C, order)
            sorted_ev = torch.cuda.Event()
            sorted_ev.record(self._side)
            host.copy_(y_C, non_blocking=True)
            arrived = torch.cuda.Event()
            arrived.record(self._side)
        C.record_stream(self._side)
        return order, y_C, sorted_ev, arrived, host

    def _encode_geometry(self, arrived, host_C, stride, postfix):
        torch.cuda.set_device(self._side.device)                 # the current device is per thread
        arrived.synchronize()
        self.coordinate_coder.encode(host_C.numpy()[:, 1:] // stride, postfix=postfix)

    def _decode_geometry(self, postfix, dev, stream):
        """`_C.bin` -> sorted stride-8 coordinate level on `dev` (coder.py:94-99: host argsort there, device sort here).
        Runs on the helper thread but enqueues on the CALLER's stream (current device and stream are per thread)."""
        torch.cuda.set_device(dev)
        path = self.coordinate_coder._path(postfix)
        if gpcc.is_native_stream(path):
            xyz8 = ops.oct_decode(_slurp(path))                  # int32 [n, 3], straight from the library
        else:
            xyz8 = np.asarray(self.coordinate_coder.decode(postfix), dtype=np.int32)
        return self._stage_geometry(xyz8, dev, stream)

    def _decode_buffers(self, C, rows=0):
        """pinned staging buffers of decode(): symbols int16 [cap, C] and coordinate level int32 [cap, 4] -> their numpy views.
        Waits for the uploads of the previous decode before handing them out again (normally long complete)."""
        self._pinned_level(max(rows, 1))
        sym = getattr(self, '_pinned_sym', None)
        if sym is None or sym.shape[0] < self._pinned_dec.shape[0] or sym.shape[1] != C:
            self._pinned_sym = torch.empty((self._pinned_dec.shape[0], C), dtype=torch.int16, pin_memory=True)
            self._pinned_sym_np = self._pinned_sym.numpy()
        return self._pinned_sym_np, self._pinned_dec_np

    def _mark_uploads(self, dev):
        self._pinned_dec_busy = torch.cuda.Event()
        self._pinned_dec_busy.record(torch.cuda.current_stream(dev))

    def _pinned_level(self, n):
        """numpy view [>= n, 4] int32 of the pinned staging buffer of the decoded coordinate level"""
        busy = getattr(self, '_pinned_dec_busy', None)
        if busy is not None:                                     # the previous upload from this buffer (normally long complete)
            busy.synchronize()
            self._pinned_dec_busy = None
        pin = getattr(self, '_pinned_dec', None)
        if pin is None or pin.shape[0] < n:
            pin = self._pinned_dec = torch.empty((max(n, 1 << 15), 4), dtype=torch.int32, pin_memory=True)
            self._pinned_dec_np = pin.numpy()
        return self._pinned_dec_np

    def _upload_level(self, n, dev):
        """asynchronous copy of the first n rows of the pinned staging buffer (current stream); the buffer is not reused before it is done"""
        y_C = self._pinned_dec[:n].to(dev, non_blocking=True)
        self._pinned_dec_busy = torch.cuda.Event()
        self._pinned_dec_busy.record(torch.cuda.current_stream(dev))
        return y_C

    def _stage_level(self, n, dev, stream):
        """the coordinate level the library has written into the pinned buffer — sorted, batch column and tensor stride in place
        (pcgc_items_decode, coord_layout 1) -> device level + the coordinate-only part of the first decoder stage.  One asynchronous
        copy; no device sort."""
        if stream == torch.cuda.current_stream(dev):                        # (the decode path: no stream switch to pay for)
            lvl8 = CoordMap(self._upload_level(n, dev), 8, unique=True)
            if n:
                lvl8.prepare_up()
            return lvl8
        with torch.cuda.stream(stream):
            lvl8 = CoordMap(self._upload_level(n, dev), 8, unique=True)
            if n:
                lvl8.prepare_up()
        return lvl8

    def _stage_geometry(self, xyz8, dev, stream):
        """decoded stride-8 voxels (host, any order) -> the sorted coordinate level on the device + the coordinate-only part of the
        first decoder stage (children level, kernel maps)"""
        n = len(xyz8)
        # batch column 0, coordinates back at tensor stride 8, assembled in pinned memory: the upload is one asynchronous copy
        host = self._pinned_level(n)[:n]
        host[:, 0] = 0
        np.multiply(xyz8, 8, out=host[:, 1:])
        with torch.cuda.stream(stream):
            y_C = self._upload_level(n, dev)
            lvl8 = CoordMap(ops.gather_coords(y_C, ops.sort_zyx(y_C)), 8, unique=True)
            if len(lvl8):
                lvl8.prepare_up()
        return lvl8

    @torch.no_grad()
    def decode(self, rho=1, postfix=''):
        """coder.py:93-112: reads the four files, returns the decoded stride-1 sparse tensor."""
        dev = require_gpu(next(self.model.decoder.parameters()).device)
        with torch.cuda.device(dev):
            return self._decode(rho, postfix, dev)

    def _decode(self, rho, postfix, dev):
        # coder.py:93-112.  The two bitstreams are independent and decoded side by side; the general path (tmc3 coordinates, or
        # NATIVE_ITEMS off) does it with a helper thread for the geometry, the native path inside one library call.
        stream = torch.cuda.current_stream(dev)
        lvl8 = None
        if self._native_items():
            # ONE library call: sizes, coordinate stream and feature stream (two native tasks side by side, each with its own pool of
            # group / segment threads — no Python thread hop on the path to the first decoder kernel), written into pinned buffers this
            # coder keeps: the symbols and the sorted coordinate level go up as two asynchronous copies.  (Measured against the
            # feature stream on a Python helper thread with this thread decoding and staging the coordinates meanwhile: the host
            # timeline looks 0.1 ms shorter that way, the frame is 0.07 ms LONGER — four A/B pairs on one box.)
            C = self.feature_coder.entropy_model._channels
            stem = self.filename + postfix
            packed = self.feature_coder.entropy_model._host_packed()
            while True:
                sym_np, level_np = self._decode_buffers(C)
                # (round 4: in two halves — the call returns once the coordinate level is decoded; the feature stream, the longer of the
                #  two, keeps decoding on the library's threads while this thread uploads the level and enqueues the coordinate-only kernels
                #  of the first decoder stage: hash, kernel map, children level and its map)
                n8, rng, counts, native = ops.frame_decode_begin(stem, C, packed, sym_np, level_np, use_sidecar=bool(INDEX_SEGMENTS), level_scale=8)
                if rng is not None:
                    break
                self._decode_buffers(C, rows=n8)                  # (a larger cloud than any before: grow and decode)
            n4, n2, n1 = counts
            if not FRAME_SPLIT:
                ops.frame_decode_end()
            if TIMELINE is not None:
                TIMELINE.append(('dec_gpu_first', time.perf_counter()))     # the first device work of the decode (level upload) is enqueued next
            try:
                if native:
                    lvl8 = self._stage_level(n8, dev, stream)
            finally:
                ops.frame_decode_end()                           # the symbols are in the pinned buffer now (or the stream's error is raised)
            sym_d = self._pinned_sym[:n8].to(dev, non_blocking=True)          # (`stream` is this thread's current stream)
            self._mark_uploads(dev)
            if not native:                                       # tmc3 stream: the subprocess protocol (helper thread in the general path)
                lvl8 = self._decode_geometry(postfix, dev, stream)
            y_F = ops.desymbolize(sym_d, rng[0])
        else:
            pending = _POOL.submit(self._decode_geometry, postfix, dev, stream)
            n4, n2, n1 = _COUNTS.unpack(_slurp(self.filename + postfix + '_num_points.bin')[:_COUNTS.size])
            y_F = self.feature_coder.decode(postfix=postfix, device=dev)
            lvl8 = pending.result()
        if min(n4, n2, n1) < 0:
            raise ops.PcgcError(f'{self.filename + postfix}_num_points.bin: negative point counts {n4, n2, n1}')
        y = SparseTensor(features=y_F, coordinate_map=lvl8)
        budgets = [[n4], [n2], [int(rho * n1)]]                  # coder.py:105-108
        _, out = self.model.decoder(y, nums_list=budgets, ground_truth_list=[None] * 3, training=False)
        return out


    # ---- collated batches: several clouds through ONE encoder / decoder pass ------------------------------------------------------
    # The reference codes one cloud per call (coder.py:80-112); its network code is batched all the same (sparse_collate,
    # data_utils.py:107; istopk per batch item, :77-89).  Small clouds — the octant blocks of BASELINE config 5, ~117 k points each —
    # leave the GPU to launch latency: ~140 kernels per block on levels of 3-40 k rows.  Collated, B clouds are one launch sequence on
    # B-times larger levels.  Results per item are IDENTICAL to coding it alone: the batch index is part of every coordinate key (no
    # kernel-map entry crosses items), every level of the batch is the concatenation of the items' levels (canonical orders are
    # first-occurrence orders of an item-contiguous input), and each output row's arithmetic chain depends on its own neighbours only.
    @torch.no_grad()
    def encode_batch(self, x, postfixes):
        """x: collated sparse tensor (item index in column 0, items contiguous); postfixes[b]: file postfix of item b.  Writes the
        four files (+ sidecar) of every item, byte-identical to encode() of the item alone; returns the sorted latent of the batch
        (items contiguous, each in (z, y, x) order)."""
        with torch.cuda.device(x.device):
            return self._encode_batch(x, list(postfixes))

    def _encode_batch(self, x, postfixes):
        B = len(postfixes)
        _check_batch_size(B, 'encode_batch')
        x = self._ingest(x)
        lvl8 = x.cmap.build_pyramid(3)
        l2 = x.cmap._down[0]
        l4 = l2._down[0]
        rows1, rows2, rows4, rows8 = x.cmap.batch_rows, l2.batch_rows, l4.batch_rows, lvl8.batch_rows
        if not (len(rows1) == len(rows2) == len(rows4) == len(rows8) == B):
            raise ValueError(f'encode_batch: {B} postfixes for a batch of {len(rows1)} items')
        y_list = self.model.encoder(x)
        order = ops.sort_zyx(lvl8.C, batch_major=True)
        y_C = ops.gather_coords(lvl8.C, order)
        y_F = ops.gather_feats(y_list[0].F, order)
        ranges, sym_h = ops.quantize_symbols_segments(y_F, rows8)             # one synchronising copy: every item's range + symbols
        host_C = y_C.cpu().numpy()
        dev = x.device
        offs = np.concatenate([[0], np.cumsum(rows8)])

        if self._native_items():
            # every item's table, range coder, sidecar, octree stream and files on native threads (one library call)
            ops.items_encode([self.filename + p for p in postfixes], sym_h, host_C[:, 1:] // lvl8.stride, rows8, ranges,
                             list(zip(rows4, rows2, rows1)), self.feature_coder.entropy_model._host_packed(), INDEX_SEGMENTS)
        else:
            def one(b):
                a, e = int(offs[b]), int(offs[b + 1])
                _dump(self.filename + postfixes[b] + '_num_points.bin', _COUNTS.pack(rows4[b], rows2[b], rows1[b]))
                self.feature_coder.encode_symbols(sym_h[a:e], ranges[b][0], ranges[b][1], postfix=postfixes[b], device=dev)
                self.coordinate_coder.encode(host_C[a:e, 1:] // lvl8.stride, postfix=postfixes[b])
            list(_batch_pool().map(one, range(B)))
        cmap = CoordMap(y_C, lvl8.stride, unique=True)
        cmap._batch_rows = list(rows8)
        return SparseTensor(y_F, coordinate_map=cmap)

    def _native_items(self):
        """the per-item host stages can run in the library (pcgc_items_*): the table is the reference-arithmetic one and the
        coordinates use the native octree stream (no tmc3 installed)"""
        return NATIVE_ITEMS and self.feature_coder.entropy_model.table_mode == 'reference' and gpcc.tmc3_path() is None

    @torch.no_grad()
    def decode_batch(self, postfixes, rho=1):
        """Reads the files of every item and decodes them in ONE decoder pass -> list of sparse tensors (batch column 0 each), equal
        to decode() of the item alone."""
        dev = require_gpu(next(self.model.decoder.parameters()).device)
        with torch.cuda.device(dev):
            return self._decode_batch(list(postfixes), rho, dev)

    def _decode_batch(self, postfixes, rho, dev):
        B = len(postfixes)
        _check_batch_size(B, 'decode_batch')

        items = None
        if self._native_items():
            stems = [self.filename + p for p in postfixes]
            rows, C, ranges, counts, native = ops.items_probe(stems)
            if C != self.feature_coder.entropy_model._channels:
                # (a damaged or foreign `_H.bin`: the library would size the table and the symbol rows from the header's channel count)
                raise ops.PcgcError(f'decode_batch: the `_H.bin` files name {C} channels, the model codes '
                                    f'{self.feature_coder.entropy_model._channels}')
            if native.all():
                # the batch's coordinate level comes out of the library sorted (items contiguous, each in its coded (z, y, x) order) and
                # with the item index in column 0
                sym_all, level = ops.items_decode(stems, rows, C, ranges, native, self.feature_coder.entropy_model._host_packed(),
                                                  use_sidecar=bool(INDEX_SEGMENTS), level_scale=8, level_out=self._pinned_level(int(rows.sum())))
                offs = np.concatenate([[0], np.cumsum(rows)])
                items = [(None, tuple(int(v) for v in counts[b]), sym_all[offs[b]:offs[b + 1]], np.float32(ranges[b, 0])) for b in range(B)]
                rows8 = [int(r) for r in rows]
                y_C = self._upload_level(int(rows.sum()), dev)
        if items is None:
            def one(b):
                xyz8 = np.asarray(self.coordinate_coder.decode(postfixes[b]), dtype=np.int32)
                counts = _COUNTS.unpack(_slurp(self.filename + postfixes[b] + '_num_points.bin')[:_COUNTS.size])
                sym_h, min_v = self.feature_coder.decode_symbols(postfix=postfixes[b], device=dev)
                if len(sym_h) != len(xyz8):
                    raise ValueError(f'item {b}: {len(xyz8)} coordinates but {len(sym_h)} latent rows')
                return xyz8, counts, sym_h, min_v
            items = list(_batch_pool().map(one, range(B)))
        if items[0][0] is not None:                                            # (items decoded one by one: assemble and sort here)
            rows8 = [len(it[0]) for it in items]
            C4 = np.zeros((sum(rows8), 4), dtype=np.int32)
            off = 0
            for b, (xyz8, _, _, _) in enumerate(items):
                C4[off:off + rows8[b], 0] = b
                C4[off:off + rows8[b], 1:] = xyz8 * 8
                off += rows8[b]
            y_C = torch.from_numpy(C4).to(dev)
            y_C = ops.gather_coords(y_C, ops.sort_zyx(y_C, batch_major=True))  # items stay contiguous, each in its coded (z, y, x) order
        sym = torch.from_numpy(np.concatenate([it[2] for it in items], 0)).to(dev)
        y_F = torch.empty(sym.shape, dtype=torch.float32, device=dev)
        off = 0
        for b, (_, _, _, min_v) in enumerate(items):                           # every item has its own symbol offset
            y_F[off:off + rows8[b]] = ops.desymbolize(sym[off:off + rows8[b]], min_v)
            off += rows8[b]
        lvl8 = CoordMap(y_C, 8, unique=True)
        lvl8._batch_rows = rows8
        y = SparseTensor(features=y_F, coordinate_map=lvl8)
        budgets = [[it[1][0] for it in items], [it[1][1] for it in items], [int(rho * it[1][2]) for it in items]]     # coder.py:105-108
        _, out = self.model.decoder(y, nums_list=budgets, ground_truth_list=[None] * 3, training=False)
        outs, off = [], 0
        for b, r in enumerate(out.cmap.batch_rows):
            c = out.C[off:off + r].clone()
            c[:, 0] = 0                                                        # coded alone, the item is batch 0
            outs.append(_coords_only(c, out.cmap.stride))
            off += r
        return outs


MAX_BATCH_ITEMS = 16            # the batch index is a 4-bit field of every coordinate key (csrc/pcgc_common.h); per-item top-k segments likewise


def _check_batch_size(n_items, what):
    if not 1 <= n_items <= MAX_BATCH_ITEMS:
        raise ops.PcgcError(f'{what}: {n_items} items; a collated batch holds 1 to {MAX_BATCH_ITEMS} clouds (code larger sets in several batches)')


def _coords_only(coords, stride):
    """decoded cloud: coordinates with unit features (what decode() of a single cloud hands on: its features are never read)"""
    return SparseTensor(lambda: torch.ones((coords.shape[0], 1), dtype=torch.float32, device=coords.device),
                        coordinate_map=CoordMap(coords, stride, unique=True))


_BATCH_POOL = None


def _batch_pool():
    """host threads for the per-item entropy / coordinate coding of a batch (native calls: they release the GIL)"""
    global _BATCH_POOL
    if _BATCH_POOL is None:
        import pcgcv2_amd
        _BATCH_POOL = ThreadPoolExecutor(max_workers=max(2, min(16, pcgcv2_amd.effective_cpus())), thread_name_prefix='pcgc-item')
    return _BATCH_POOL


# ------------------------------------------------------------------------------------------------ CLI (coder.py:114-184)
def _parse_cli(argv):
    import argparse
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--ckptdir", default='ckpts/r3_0.10bpp.pth')
    p.add_argument("--filedir", default='../../../testdata/8iVFB/longdress_vox10_1300.ply')
    p.add_argument("--scaling_factor", type=float, default=1.0, help='scaling_factor')
    p.add_argument("--rho", type=float, default=1.0, help='the ratio of the number of output points to the number of input points')
    p.add_argument("--res", type=int, default=1024, help='resolution')
    p.add_argument("--outdir", default='./output')
    return p.parse_args(argv)


class _Stopwatch:
    def __init__(self, label, digits=3, sync=True):
        self.label, self.digits, self.sync = label, digits, sync

    def __enter__(self):
        self.t0 = time.time()
        return self

    def __exit__(self, *exc):
        if self.sync and torch.cuda.is_available():
            torch.cuda.synchronize()
        print(f'{self.label}:\t', round(time.time() - self.t0, self.digits), 's')


def main(argv=None):
    args = _parse_cli(argv)
    with _Stopwatch('Loading Time', 4, sync=False):
        x = load_sparse_tensor(args.filedir, device)
    os.makedirs(args.outdir, exist_ok=True)
    prefix = os.path.join(args.outdir, os.path.split(args.filedir)[-1].split('.')[0])
    print(prefix)

    print('=' * 10, 'Test', '=' * 10)
    if not os.path.exists(args.ckptdir):
        raise FileNotFoundError(args.ckptdir)
    model = PCCModel().to(device)
    model.load_state_dict(torch.load(args.ckptdir, map_location=device)['model'])
    print('load checkpoint from \t', args.ckptdir)

    coder = Coder(model=model, filename=prefix)
    scaled = args.scaling_factor != 1
    x_in = scale_sparse_tensor(x, factor=args.scaling_factor) if scaled else x
    with _Stopwatch('Enc Time'):
        coder.encode(x_in)
    with _Stopwatch('Dec Time'):
        x_dec = coder.decode(rho=args.rho)
    if scaled:
        x_dec = scale_sparse_tensor(x_dec, factor=1.0 / args.scaling_factor)

    bits = stream_bits(prefix)
    bpps = (bits / len(x)).round(3)                      # per-file rounding, then summed (coder.py:171-173)
    print('bits:\t', bits, '\nbpps:\t', bpps)
    print('bits:\t', sum(bits), '\nbpps:\t', sum(bpps).round(3))

    with _Stopwatch('Write PC Time', sync=False):
        write_ply_ascii_geo(prefix + '_dec.ply', x_dec.C.detach().cpu().numpy()[:, 1:])
    with _Stopwatch('PC Error Metric Time', sync=False):
        metrics = pc_error(args.filedir, prefix + '_dec.ply', res=args.res, show=False)
    print('D1 PSNR:\t', metrics["mseF,PSNR (p2point)"][0])


if __name__ == '__main__':
    main()

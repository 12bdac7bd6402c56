"""Factorized entropy bottleneck (reference entropy_model.py:42-196), inference path only.

Same parameter names as the reference — `_matrices.{0..3}`, `_biases.{0..3}`, `_factors.{0..3}` and the three aliases
`matrix` / `bias` / `factor` that the reference creates by assigning `self.matrix = Parameter(...)` inside its
constructor loop (entropy_model.py:68-80) — so a strict load_state_dict of a reference checkpoint succeeds.

compress/decompress: quantisation, symbol range and symbolisation run on the GPU (pcgc_round_minmax, pcgc_symbolize);
only int16 symbols cross to the host, where the sequential range coder (pcgc_rc_encode / pcgc_rc_decode,
torchac-compatible) runs.  The reference's [N8, 8, L+1] fp32 CDF expansion (entropy_model.py:173) never exists.

The 8 x (L+1) CDF table decides every bit of `_F.bin`, so encoder and decoder — here or in the reference — must derive
the same uint16 values.  `table_mode`:
  'reference' (default)  the table is evaluated on the host with the reference's own arithmetic: the same torch-CPU fp32
                         operator sequence on tensors of the same shape and layout as entropy_model.py:82-101,112-130,
                         142-149, then torchac's published 16-bit normalisation — issued from C++ (libpcgc_reftable.so,
                         csrc/reftable.cpp); 'reference-python' is the same sequence from Python (reference_table).  Pinned bit for bit to golden tables
                         generated from the reference (tests/golden/entropy_tables.npz): a stream written here decodes
                         in the reference on the same host and vice versa.  The table is consumed by the host range
                         coder anyway, so nothing extra crosses PCIe.
  'device'               the fused HIP kernel pcgc_cdf_table (fp64 evaluation rounded to fp32): self-consistent between
                         this encoder and decoder, within 1 count of the reference table, NOT interoperable with it.
"""
import threading
import weakref

import numpy as np
import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from . import ops
from ._lib import PcgcError


_TABLE_LOCK = threading.Lock()
TABLE_CACHE = True          # host CDF tables are kept per (parameters, range); table_cache(False) = evaluate on every call, as the reference does


def table_cache(on=None, clear=False):
    """Policy of the CDF-table caches (this module's and the library's, pcgc_table_cache).  `clear`: drop what is cached now — the next
    compress / decompress of every (parameters, range) evaluates its table again, which is what the reference does on EVERY call
    (entropy_model.py:165-171, 185-190) and what a decoder process that did not encode the frame sees; `on` = False / True: stop /
    resume caching.  -> tables dropped from the library's cache."""
    global TABLE_CACHE
    from ._lib import lib
    dropped = 0
    if on is not None:
        TABLE_CACHE = bool(on)
        dropped += lib().pcgc_table_cache(1 if on else -1)
    if clear:
        dropped += lib().pcgc_table_cache(0)
        from ._lib import reftable_lib
        reftable_lib().pcgc_reference_table_clear()          # ... and the parameter-only operators (softplus / tanh) kept per parameter set
        with _TABLE_LOCK:
            for m in list(_MODELS):
                m.__dict__.pop('_table_cache', None)
    return dropped


_MODELS = weakref.WeakSet()


class EntropyBottleneck(nn.Module):
    def __init__(self, channels, init_scale=8, filters=(3, 3, 3)):
        super().__init__()
        self._likelihood_bound = 1e-9
        self._init_scale = float(init_scale)
        self._filters = tuple(int(f) for f in filters)
        self._channels = channels
        if self._filters != (3, 3, 3):
            raise NotImplementedError('the HIP CDF kernel is specialised to filters=(3,3,3) (pcc_model.py:13 uses the default)')
        filt = (1,) + self._filters + (1,)
        scale = self._init_scale ** (1 / (len(self._filters) + 1))
        self._matrices, self._biases, self._factors = nn.ParameterList(), nn.ParameterList(), nn.ParameterList()
        for i in range(len(self._filters) + 1):
            m = Parameter(torch.full((channels, filt[i + 1], filt[i]), float(np.log(np.expm1(1.0 / scale / filt[i + 1])))))
            b = Parameter(torch.from_numpy(np.random.uniform(-0.5, 0.5, (channels, filt[i + 1], 1)).astype(np.float32)))
            f = Parameter(torch.zeros(channels, filt[i + 1], 1))
            self._matrices.append(m); self._biases.append(b); self._factors.append(f)
        # the reference's accidental aliases of the LAST layer's tensors (same Parameter objects)
        self.matrix, self.bias, self.factor = self._matrices[-1], self._biases[-1], self._factors[-1]
        self.table_mode = 'reference'
        self._packed = self._packed_stamp = None
        self._host = self._host_stamp = None
        _MODELS.add(self)                                    # (table_cache(clear=True) reaches every live model's cache)

    def cpu(self):
        """coder.py:44 calls `entropy_model.cpu()`; the tables are evaluated on the GPU here, so the module stays put."""
        return self

    def _stamp(self):
        """identity + version of the 12 parameter tensors: any in-place update, re-assignment, .to() or load_state_dict —
        through this module or any parent container — changes it, so derived copies can never go stale."""
        return tuple((p.data_ptr(), p._version, p.device) for lst in (self._matrices, self._biases, self._factors)
                     for p in lst._parameters.values())

    def packed_params(self, device):
        """352 fp32: matrices 0..3 | biases 0..3 | factors 0..3 — the layout pcgc_cdf_table expects."""
        stamp = (self._stamp(), device)
        if self._packed is None or self._packed_stamp != stamp:
            parts = [p.detach().reshape(-1).float() for lst in (self._matrices, self._biases, self._factors) for p in lst]
            self._packed, self._packed_stamp = torch.cat(parts).to(device).contiguous(), stamp
        return self._packed

    def invalidate(self):
        self._packed = self._host = self._hpacked = None
        self.__dict__.pop('_table_cache', None)

    def _host_params(self):
        """fp32 CPU copies of (matrices, biases, factors), refreshed when the parameters change."""
        stamp = self._stamp()
        if self._host is None or self._host_stamp != stamp:
            mats, biases, factors = ([p.detach().to('cpu', torch.float32) for p in lst]
                                     for lst in (self._matrices, self._biases, self._factors))
            # the parameter-only terms of _logits_cumulative — softplus(matrix) (entropy_model.py:94) and tanh(factor) (:97) —
            # are the same tensors in every call: evaluated once per parameter set with the same operators
            self._host = ([torch.nn.functional.softplus(m) for m in mats], biases, [torch.tanh(f) for f in factors])
            self._host_stamp = stamp
        return self._host

    @torch.no_grad()
    def reference_table(self, min_v, max_v):
        """The CDF table exactly as the reference derives it on the CPU -> (cdf fp32 [C, L+1], table uint16 ndarray [C, L+1]).

        Same torch-CPU fp32 operators, in the same order, on tensors of the same shape and memory layout as
        entropy_model.py:163-172 / 181-189 (symbols -> _likelihood :112-130 -> _logits_cumulative :82-101 -> clamp ->
        _pmf_to_cdf :142-149): operator order, in-place forms and layouts are part of the contract, because torch's CPU
        kernels pick vectorised / scalar-tail / BLAS paths by shape.  The last step is torchac 0.9.3's published
        `_convert_to_int_and_normalize` (scale by 2^16 - (Lp - 1), round, int16, + arange(Lp))."""
        sp_mats, biases, th_factors = self._host_params()
        C = self._channels
        sym = torch.arange(float(min_v), float(max_v) + 1)                    # float32, like arange(min_v, max_v + 1) there
        pts = sym.reshape(-1, 1).repeat(1, C)                                 # [L, C]
        grid = pts.permute(1, 0).contiguous()
        shape = grid.size()
        grid = grid.view(shape[0], 1, -1)                                     # [C, 1, L]
        ends = []
        for half in (-0.5, 0.5):                                              # lower = f(v - 0.5), upper = f(v + 0.5)
            z = grid + half
            for m, b, f in zip(sp_mats, biases, th_factors):
                z = torch.matmul(m, z)
                z += b
                z += f * torch.tanh(z)
            ends.append(z)
        lower, upper = ends
        sign = -torch.sign(torch.add(lower, upper))
        lik = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
        lik = lik.view(shape).permute(1, 0)                                   # [L, C] view, as _likelihood returns it
        pmf = torch.clamp(lik, min=self._likelihood_bound).permute(1, 0)      # [C, L]
        cdf = pmf.cumsum(dim=-1)
        cdf = torch.cat([torch.zeros(pmf.shape[:-1] + (1,), dtype=pmf.dtype), cdf], dim=-1).clamp(max=1.)
        return cdf, self.convert_to_int_and_normalize(cdf)

    @staticmethod
    def convert_to_int_and_normalize(cdf):
        """torchac 0.9.3's published `_convert_to_int_and_normalize` (needs_normalization=True) on an fp32 CPU tensor [.., Lp]:
        cdf_float.mul(2^16 - (Lp - 1)).round().to(int16).add_(arange(Lp, int16)) -> uint16 ndarray."""
        Lp = cdf.shape[-1]
        top = torch.tensor(2, dtype=torch.float32).pow_(16) - (Lp - 1)
        q = cdf.mul(top).round().to(dtype=torch.int16)
        q.add_(torch.arange(Lp, dtype=torch.int16))
        return q.contiguous().numpy().view(np.uint16)

    def cdf_table(self, min_v, max_v, device):
        """device-kernel table (table_mode 'device'): (uint16 bit patterns as int16 tensor [C, L+1], fp32 cdf)."""
        q, f = ops.cdf_table(self.packed_params(device), self._channels, float(min_v), float(max_v))
        return q, f

    def _host_packed(self):
        """the 44*C parameters as one contiguous fp32 CPU array (matrices | biases | factors), refreshed when they change"""
        stamp = self._stamp()
        if getattr(self, '_hpacked', None) is None or self._hpacked_stamp != stamp:
            parts = [p.detach().reshape(-1).to('cpu', torch.float32) for lst in (self._matrices, self._biases, self._factors)
                     for p in lst._parameters.values()]
            self._hpacked, self._hpacked_stamp = torch.cat(parts).contiguous().numpy(), stamp
        return self._hpacked

    def reference_table_native(self, min_v, max_v, want_cdf=False):
        """reference_table through libpcgc_reftable.so: the same ATen operator sequence issued from C++ (csrc/reftable.cpp), without
        ~60 Python dispatches on the critical path of every encode and decode.  -> uint16 ndarray [C, L+1] (and the fp32 cdf)."""
        from ._lib import reftable_lib
        P = self._host_packed()
        L = int(np.float32(max_v) - np.float32(min_v)) + 1
        q = np.empty((self._channels, L + 1), np.uint16)
        cdf = np.empty((self._channels, L + 1), np.float32) if want_cdf else None
        rc = reftable_lib().pcgc_reference_table(P.ctypes.data, self._channels, float(min_v), float(max_v), q.ctypes.data,
                                                 None if cdf is None else cdf.ctypes.data)
        if rc != 0:
            raise PcgcError(f'pcgc_reference_table failed ({rc})')
        return (q, cdf) if want_cdf else q

    TABLE_CACHE_SIZE = 16

    def host_table(self, min_v, max_v, device, want_crc=False):
        """uint16 ndarray [C, L+1] on the host, by the configured table_mode (read-only: cached).

        The table is a pure function of (parameters, min_v, max_v, table_mode): the last TABLE_CACHE_SIZE are kept, keyed by the
        parameters' stamp — the decode of a frame this process has just encoded, every repeat of a frame, and every frame of a
        sequence whose latent range repeats re-use the evaluated table (0.1-0.2 ms of ATen operator dispatch each) and its
        CRC-32; results are identical by construction (same stamp = same parameter values)."""
        import zlib
        if self.table_mode not in ('reference', 'reference-python', 'device'):
            raise PcgcError(f"table_mode must be 'reference', 'reference-python' or 'device', got {self.table_mode!r}")
        if self.table_mode == 'device' and device is None:
            device = self._matrices[0].device                                    # (encode_symbols / decode_symbols may not name one)
        key = (self._stamp(), float(min_v), float(max_v), self.table_mode, str(device) if self.table_mode == 'device' else '')
        with _TABLE_LOCK:                                                        # (compress_symbols / decompress_symbols run on pool threads)
            cache = self.__dict__.setdefault('_table_cache', {})
            hit = cache.get(key) if TABLE_CACHE else None
        if hit is None:
            if self.table_mode == 'reference':
                table = self.reference_table_native(min_v, max_v)
            elif self.table_mode == 'reference-python':
                table = self.reference_table(min_v, max_v)[1]
            else:
                table = self.cdf_table(min_v, max_v, device)[0].cpu().numpy().view(np.uint16)
            table = np.ascontiguousarray(table)
            table.setflags(write=False)
            hit = (table, zlib.crc32(table.tobytes()))
            if TABLE_CACHE:
                with _TABLE_LOCK:
                    while len(cache) >= self.TABLE_CACHE_SIZE:
                        cache.pop(next(iter(cache)), None)                       # (dicts keep insertion order: drop the oldest)
                    cache[key] = hit
        return hit if want_crc else hit[0]

    @torch.no_grad()
    def compress(self, inputs, checkpoints=0, info=None):
        """entropy_model.py:151-176 -> (bytes, min_v ndarray[1], max_v ndarray[1]); with checkpoints > 0 a fourth element: the
        decoding index of ops.rc_encode (decoder states at that many row boundaries; the bytes are the same either way).
        `info` (optional dict) receives 'table_crc': CRC-32 of the uint16 table the stream was coded with (the guard a decoder
        on another host can check, see decompress)."""
        if inputs.dim() != 2 or inputs.shape[1] != self._channels:
            raise PcgcError(f'compress expects [N, {self._channels}] features')
        # symbol range + symbols in one enqueue and ONE synchronising copy; the table is then evaluated (or found in the cache) on
        # the host, where the range coder consumes it
        min_v, max_v, sym_h = ops.quantize_symbols(inputs)
        return self.compress_symbols(sym_h, min_v, max_v, checkpoints=checkpoints, info=info, device=inputs.device)

    def compress_symbols(self, sym_h, min_v, max_v, checkpoints=0, info=None, device=None):
        """The host half of compress(): int16 symbols [N, C] (= round(x) - min_v) + their range -> the same tuple compress returns.
        Thread-safe: the items of a batch are range-coded side by side (coder.Coder.encode_batch)."""
        table_h, crc = self.host_table(min_v, max_v, device, want_crc=True)
        if info is not None:
            info['table_crc'] = crc
        if checkpoints > 0:
            strings, index = ops.rc_encode(table_h, sym_h, checkpoints=checkpoints)
            return strings, np.array([min_v], np.float32), np.array([max_v], np.float32), index
        strings = ops.rc_encode(table_h, sym_h)
        return strings, np.array([min_v], np.float32), np.array([max_v], np.float32)

    @torch.no_grad()
    def decompress(self, strings, min_v, max_v, shape, channels, device=None, on_table_launched=None, index=None, expect_table_crc=None):
        """entropy_model.py:178-196 -> fp32 [shape[0], channels] on `device`.  `on_table_launched` (optional) is called
        before this thread starts on the table: the place to start concurrent host work.  `index` (optional): the decoding
        index compress(..., checkpoints=k) returned for these bytes -> the segments are decoded in parallel.
        `expect_table_crc` (optional): CRC-32 of the ENCODER's table; if this host derives another table (torch-CPU kernels differ
        between CPU kinds and torch builds by a count here and there) the stream would decode to garbage — raise instead."""
        device = torch.device('cuda') if device is None else device
        sym_h, min_v = self.decompress_symbols(strings, min_v, max_v, shape, channels, device=device, on_table_launched=on_table_launched,
                                               index=index, expect_table_crc=expect_table_crc)
        sym = torch.from_numpy(sym_h).to(device)
        return ops.desymbolize(sym, min_v)

    def decompress_symbols(self, strings, min_v, max_v, shape, channels, device=None, on_table_launched=None, index=None, expect_table_crc=None):
        """The host half of decompress(): -> (int16 symbols ndarray [shape[0], channels], min_v); values = symbols + min_v.
        Thread-safe (coder.Coder.decode_batch decodes the items of a batch side by side)."""
        min_v, max_v = np.float32(np.asarray(min_v).reshape(-1)[0]), np.float32(np.asarray(max_v).reshape(-1)[0])
        if on_table_launched is not None:
            on_table_launched()
        table_h, crc = self.host_table(min_v, max_v, device, want_crc=True)
        if expect_table_crc is not None and int(expect_table_crc) != crc:
            raise PcgcError(f'the CDF table derived on this host (CRC-32 {crc:08x}, table_mode {self.table_mode!r}) is not the one this '
                            f'stream was coded with ({int(expect_table_crc):08x}): other host CPU kind / torch build / checkpoint. '
                            'Decoding would return noise; decode where the stream was encoded, or re-encode with table_mode="device".')
        n = int(shape[0]) * int(channels)
        sym_h = ops.rc_decode(table_h, strings, n, index=index)
        return sym_h.reshape(int(shape[0]), int(channels)), min_v

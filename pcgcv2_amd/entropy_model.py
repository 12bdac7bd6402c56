"""Factorized entropy bottleneck (reference entropy_model.py:42-196), inference path only.

Same parameter names as the reference — `_matrices.{0..3}`, `_biases.{0..3}`, `_factors.{0..3}` and the three aliases
`matrix` / `bias` / `factor` that the reference creates by assigning `self.matrix = Parameter(...)` inside its
constructor loop (entropy_model.py:68-80) — so a strict load_state_dict of a reference checkpoint succeeds.

compress/decompress: quantisation, symbol range and the CDF table run on the GPU (pcgc_round_minmax, pcgc_symbolize,
pcgc_cdf_table); only int16 symbols and the 8 x (L+1) 16-bit table cross to the host, where the sequential range coder
(pcgc_rc_encode / pcgc_rc_decode, torchac-compatible) runs.  The reference's [N8, 8, L+1] fp32 CDF expansion
(entropy_model.py:173) never exists.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from . import ops
from ._lib import PcgcError


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
        self._packed = None

    def cpu(self):
        """coder.py:44 calls `entropy_model.cpu()`; the tables are evaluated on the GPU here, so the module stays put."""
        return self

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def packed_params(self, device):
        """352 fp32: matrices 0..3 | biases 0..3 | factors 0..3 — the layout pcgc_cdf_table expects."""
        if self._packed is None or self._packed.device != device:
            parts = [p.detach().reshape(-1).float() for lst in (self._matrices, self._biases, self._factors) for p in lst]
            self._packed = torch.cat(parts).to(device).contiguous()
        return self._packed

    def invalidate(self):
        self._packed = None

    def cdf_table(self, min_v, max_v, device):
        q, f = ops.cdf_table(self.packed_params(device), self._channels, float(min_v), float(max_v))
        return q, f

    @torch.no_grad()
    def compress(self, inputs):
        """entropy_model.py:151-176 -> (bytes, min_v ndarray[1], max_v ndarray[1])."""
        if inputs.dim() != 2 or inputs.shape[1] != self._channels:
            raise PcgcError(f'compress expects [N, {self._channels}] features')
        prep = ops.compress_prepare(inputs, self.packed_params(inputs.device), self._channels)     # one D2H + sync
        if prep is not None:
            min_v, max_v, sym_h, table_h = prep
        else:                                                        # alphabet larger than the staged table: two-phase path
            mm = ops.round_minmax(inputs).cpu().numpy()
            min_v, max_v = np.float32(mm[0]), np.float32(mm[1])
            sym_h = ops.symbolize(inputs, min_v).cpu().numpy()
            table_h = self.cdf_table(min_v, max_v, inputs.device)[0].cpu().numpy().view(np.uint16)
        strings = ops.rc_encode(table_h, sym_h)
        return strings, np.array([min_v], np.float32), np.array([max_v], np.float32)

    @torch.no_grad()
    def decompress(self, strings, min_v, max_v, shape, channels, device=None, on_table_launched=None):
        """entropy_model.py:178-196 -> fp32 [shape[0], channels] on `device`.  `on_table_launched` (optional) is called once
        the CDF-table kernel is enqueued, before this thread blocks on its result: the place to start concurrent host work."""
        device = torch.device('cuda') if device is None else device
        min_v, max_v = np.float32(np.asarray(min_v).reshape(-1)[0]), np.float32(np.asarray(max_v).reshape(-1)[0])
        table, _ = self.cdf_table(min_v, max_v, device)
        if on_table_launched is not None:
            on_table_launched()
        n = int(shape[0]) * int(channels)
        sym_h = ops.rc_decode(table.cpu().numpy().view(np.uint16), strings, n)
        sym = torch.from_numpy(sym_h.reshape(int(shape[0]), int(channels))).to(device)
        return ops.desymbolize(sym, min_v)

"""Thin functional layer over the C-ABI (include/pcgc_hip.h): torch tensors in, torch tensors out.

PyTorch is used for device memory and streams only; all arithmetic happens in libpcgc_hip.so.  Every function
requires ROCm device tensors and raises otherwise — there is no CPU path in the product.
"""
import os as _os

import numpy as np
import torch

from ._lib import lib, check, PcgcError


def _stream(t=None):
    """HIP stream the call is enqueued on: the current stream of the OPERAND's device (not of the current device — they differ
    in a single-process multi-GPU program that never calls set_device)."""
    if t is None:
        return torch.cuda.current_stream().cuda_stream
    return torch.cuda.current_stream(t.device).cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _dev(t, dtype, what):
    if not t.is_cuda:
        raise PcgcError(f'{what}: expected a ROCm device tensor, got {t.device} (no CPU fallback exists)')
    if t.dtype != dtype:
        raise PcgcError(f'{what}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise PcgcError(f'{what}: tensor must be contiguous')
    return t


def _i32(t, what='coords'):
    return _dev(t, torch.int32, what)


def _f32(t, what='feats'):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise PcgcError(f'{what}: expected float32 device tensor, got {t.dtype} on {t.device}')
    return t


def _ld(t):
    """leading dimension (row stride in floats) of a 2-D row-major view whose rows are contiguous."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise PcgcError('feature views must be 2-D with unit column stride')
    return t.stride(0)


# ------------------------------------------------------------------------------------------------ profiling hook
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 MFMA peak
_CHILD_WEIGHT = None


def child_pairs(parent_nbr):
    """(in,out) pairs of the CHILDREN level's k3 map, counted from the parent map (the children map is never built): a present
    parent neighbour at offset kp contributes as many (child row, offset) pairs as halo cells of that parent are reached:
    64 for the centre, 16 per face, 4 per edge, 1 per corner neighbour."""
    global _CHILD_WEIGHT
    if _CHILD_WEIGHT is None or _CHILD_WEIGHT.device != parent_nbr.device:
        w = np.zeros(27, np.int64)
        for kp, jc, reach in _halo_cells():
            w[kp] += len(reach)
        _CHILD_WEIGHT = torch.from_numpy(w).to(parent_nbr.device)
    return int(((parent_nbr >= 0).sum(1) * _CHILD_WEIGHT).sum().item())


class _Profile:
    """Brackets the sparse-conv launches with HIP events on the stream they are launched on, and turns the timings into the
    `roofline` object of bench.py.  Per launch, with P = (in,out) pairs of the level's k3 map (counted, not assumed), n rows:
        gathered bytes      k3 conv: P*Cin*4 + P*8 + n*Cout*4 ;  k1 conv: n*(Cin+Cout)*4 — every (in,out) pair
                         charged: the re-use of a row by its ~18 outputs is served by L2 / LDS, so this rate can exceed the HBM peak
        compulsory bytes                   every input row, kernel-map entry and output row once: what HBM has to move
        flops                              2*P*Cin*Cout (+ 2*n*Cin*Cout for k1 parts);  k2 s2 down / generative up: P = fine rows
        mfma_issued                        fp32 MFMA flops the kernel issues (zero-padded columns and absent rows included)
    A fused InceptionResNet pass is charged the sum of the convs it computes:
        pass A: k3 C->C/4 + k1 C->C/4          pass B: k3 C/4->C/2 + k3 C/4->C/4 + k1 C/4->C/2.
    Records are per (kernel, level) key; the report groups them by kernel NAME (the text of `kernel` before ' (': what a rocprofv3
    --stats line pools too), so that `roofline.kernel` is a statistic of one compiled kernel over every level it serves."""

    TIE = 0.05                     # kernel names whose per-step time is within 5 % of the largest are tied: the lower fraction of peak wins

    def __init__(self):
        self.reset(False)

    def reset(self, enabled=False, only=None):
        self.enabled = enabled
        self.only = None if only is None else frozenset(only)      # bracket just these (kernel, level) keys: keeps the event overhead out of a timed region
        self.counting = False
        self.records = {}          # key -> dict(kernel, n, formulas, events=[(e0,e1)])
        self.pairs = {}            # n_out -> P of the level's k3 map

    def want(self, key):
        return self.enabled and (self.only is None or key in self.only)

    @staticmethod
    def name_of(kernel):
        return kernel.split(' (')[0]

    def bracket(self, key, kernel, n, gathered, flops, compulsory=None, mfma_issued=None, pairs=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r = self.records.setdefault(key, {'kernel': kernel, 'n': n, 'gathered': gathered, 'flops': flops, 'compulsory': compulsory,
                                          'mfma_issued': mfma_issued, 'pairs': pairs, 'events': []})
        r['events'].append((e0, e1))
        return e0, e1

    def count(self, nbr):
        n = nbr.shape[1]
        if n not in self.pairs:
            self.pairs[n] = int((nbr >= 0).sum().item())

    def count_children(self, parent_nbr):
        n = 8 * parent_nbr.shape[1]
        if n not in self.pairs:
            self.pairs[n] = child_pairs(parent_nbr)

    def detail(self):
        """one entry per (kernel, level): mean AND median launch time (a first-touch allocation or a preempted launch must not decide a ranking)"""
        out = []
        for key, r in self.records.items():
            ts = sorted(e0.elapsed_time(e1) for e0, e1 in r['events'])
            ms = sum(ts)
            us = ms / len(ts) * 1e3
            med = (ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])) * 1e3
            d = {'key': key, 'name': self.name_of(r['kernel']), 'kernel': r['kernel'], 'n_out': r['n'], 'launches': len(ts), 'ms': ms, 'avg_us': us, 'median_us': med}
            P = r['pairs'] if r['pairs'] is not None else self.pairs.get(r['n'])     # (k2 s2 convs carry their own P = fine rows)
            if P is not None:
                d['pairs'] = P
                d['gathered_bytes'] = r['gathered'](P)
                d['flops'] = r['flops'](P)
                d['gathered_GBps'] = d['gathered_bytes'] / (us * 1e-6) / 1e9
                d['TFLOPs'] = d['flops'] / (us * 1e-6) / 1e12
                if r['compulsory'] is not None:
                    d['compulsory_bytes'] = r['compulsory']
                    d['compulsory_GBps'] = r['compulsory'] / (us * 1e-6) / 1e9
                if r['mfma_issued'] is not None:
                    issued = r['mfma_issued'](P) if callable(r['mfma_issued']) else r['mfma_issued']      # (a packed kernel issues per present pair)
                    d['mfma_issued_flops'] = issued
                    d['mfma_issued_TFLOPs'] = issued / (us * 1e-6) / 1e12
            out.append(d)
        return sorted(out, key=lambda d: -d['ms'])

    def families(self, steps, peak_gbs):
        """the fixed report: one entry per kernel NAME, its levels pooled (sum of flops / sum of time): us per step, launches per step, fraction
        of the fp32 MFMA peak (algorithmic flops), issued / algorithmic, compulsory-traffic fraction of the HBM peak; sorted by name"""
        groups = {}
        for d in self.detail():
            if 'pairs' in d:
                groups.setdefault(d['name'], []).append(d)
        fam = []
        for name, ds in groups.items():
            ms = sum(d['ms'] for d in ds)
            med_ms = sum(d['median_us'] * d['launches'] for d in ds) * 1e-3
            fl = sum(d['flops'] * d['launches'] for d in ds)
            comp = sum(d.get('compulsory_bytes', 0) * d['launches'] for d in ds)
            issued = sum(d.get('mfma_issued_flops', 0) * d['launches'] for d in ds)
            f = {'kernel': name, 'levels': sorted({d['n_out'] for d in ds}), 'launches_per_step': round(sum(d['launches'] for d in ds) / steps, 2),
                 'us_per_step': round(ms * 1e3 / steps, 1), 'median_us_per_step': round(med_ms * 1e3 / steps, 1),
                 'TFLOPs': round(fl / (ms * 1e-3) / 1e12, 2), 'frac': round(fl / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                 'issued_over_algorithmic': round(issued / fl, 2) if issued and fl else None,
                 'compulsory_frac_of_hbm': round(comp / (ms * 1e-3) / 1e9 / peak_gbs, 4)}
            fam.append(f)
        return sorted(fam, key=lambda f: f['kernel'])

    def dominant(self, steps, peak_gbs):
        """(kernel name, its keys): the name with the largest per-step time (median launch time x launches, so one slow sample cannot decide);
        names within TIE of it are tied and the LOWEST fraction of the fp32 MFMA peak wins — a rule that returns the same kernel on every box
        as long as the ranking of the fractions holds, where "largest time" alone flipped between four groups 3 % apart."""
        fam = self.families(steps, peak_gbs)
        if not fam:
            return None, ()
        top = max(f['median_us_per_step'] for f in fam)
        tied = [f for f in fam if f['median_us_per_step'] >= (1.0 - self.TIE) * top]
        best = min(tied, key=lambda f: (f['frac'], f['kernel']))
        keys = tuple(k for k, r in self.records.items() if self.name_of(r['kernel']) == best['kernel'])
        return best['kernel'], keys

    def summary(self, peak_gbs, steps, name=None):
        """roofline of one kernel name (default: `dominant`), its levels pooled: achieved = sum of algorithmic flops / sum of launch time.
        `bound` is the roofline the kernel sits closer to: "mfma" (algorithmic fp32 flops against the dense fp32 MFMA peak) or
        "hbm" (compulsory bytes against the HBM peak).  `aggregate` = every bracketed launch; `families` = the per-name table."""
        d_all = [r for r in self.detail() if 'pairs' in r]
        if not d_all:
            return None
        if name is None:
            name, _ = self.dominant(steps, peak_gbs)
        d = [r for r in d_all if r['name'] == name]
        if not d:
            return None
        t_s = sum(r['ms'] for r in d) * 1e-3
        launches = sum(r['launches'] for r in d)
        fl = sum(r['flops'] * r['launches'] for r in d)
        comp = sum(r.get('compulsory_bytes', 0) * r['launches'] for r in d)
        gath = sum(r['gathered_bytes'] * r['launches'] for r in d)
        issued = sum(r.get('mfma_issued_flops', 0) * r['launches'] for r in d)
        tf, cg = fl / t_s / 1e12, comp / t_s / 1e9
        hbm_frac, mfma_frac = cg / peak_gbs, tf / MFMA_F32_PEAK_TFLOPS
        # an MFMA kernel is priced against the matrix peak unless its compulsory traffic rate is the larger fraction of ITS peak even
        # compared with the pipe utilisation (issued flops), i.e. unless it really is the memory side that is closer to its limit
        is_mfma = issued > 0 and max(mfma_frac, issued / t_s / 1e12 / MFMA_F32_PEAK_TFLOPS) >= hbm_frac
        roof = {'bound': 'mfma' if is_mfma else 'hbm'}
        if is_mfma:
            roof.update(achieved=round(tf, 2), peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s', frac=round(mfma_frac, 4))
        else:
            roof.update(achieved=round(cg, 2), peak=peak_gbs, unit='GB/s', frac=round(hbm_frac, 4))
        roof.update(traffic=None, kernel=name, description=d[0]['kernel'], avg_launch_us=round(t_s * 1e6 / launches, 2), launches_timed=launches,
                    launches_per_step=round(launches / steps, 2),
                    selection=f'kernel NAME with the largest per-step time (median launch time x launches, levels pooled); names within {int(self.TIE * 100)} % '
                              'are tied and the lowest fraction of peak wins',
                    levels=[{'n_out': r['n_out'], 'pairs': r['pairs'], 'launches': r['launches'], 'avg_launch_us': round(r['avg_us'], 2),
                             'flops_per_launch': r['flops'], 'frac_of_fp32_mfma_peak': round(r['TFLOPs'] / MFMA_F32_PEAK_TFLOPS, 4),
                             'compulsory_bytes_per_launch': r.get('compulsory_bytes'),
                             'mfma_issued_flops_per_launch': r.get('mfma_issued_flops')} for r in sorted(d, key=lambda r: -r['n_out'])],
                    algorithmic={'flops_per_launch': fl / launches, 'TFLOPs': round(tf, 2), 'frac_of_fp32_mfma_peak': round(mfma_frac, 4),
                                 'compulsory_bytes_per_launch': comp / launches, 'compulsory_GBps': round(cg, 1), 'frac_of_hbm_peak': round(hbm_frac, 4),
                                 'gathered_bytes_per_launch': gath / launches, 'gathered_GBps': round(gath / t_s / 1e9, 1),
                                 'note': 'per-launch figures are means over the launches of this kernel name (levels pooled: see `levels`); gathered = SURVEY 8d '
                                         'formula, every (in,out) pair charged — the row re-use is served by L2/LDS, not HBM'})
        if issued:
            roof['mfma_issued'] = {'flops_per_launch': issued / launches, 'TFLOPs': round(issued / t_s / 1e12, 2),
                                   'pipe_utilisation': round(issued / t_s / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), 'issued_over_algorithmic': round(issued / fl, 2),
                                   'note': 'fp32 MFMA flops issued (zero-padded columns and absent rows included) / time'}
        roof['all_launches'] = self.aggregate(peak_gbs, steps)
        return roof

    def aggregate(self, peak_gbs, steps):
        d = [r for r in self.detail() if 'pairs' in r]
        tot_g = sum(r['gathered_bytes'] * r['launches'] for r in d)
        tot_c = sum(r.get('compulsory_bytes', 0) * r['launches'] for r in d)
        tot_f = sum(r['flops'] * r['launches'] for r in d)
        tot_ms = sum(r['ms'] for r in d)
        return {'ms_per_step': round(tot_ms / steps, 3), 'launches_per_step': round(sum(r['launches'] for r in d) / steps, 2),
                'GFLOP_per_step': round(tot_f / steps / 1e9, 2),
                'TFLOPs': round(tot_f / (tot_ms * 1e-3) / 1e12, 2), 'frac_of_fp32_mfma_peak': round(tot_f / (tot_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                'compulsory_GBps': round(tot_c / (tot_ms * 1e-3) / 1e9, 1), 'frac_of_hbm_peak': round(tot_c / (tot_ms * 1e-3) / 1e9 / peak_gbs, 4),
                'gathered_GBps': round(tot_g / (tot_ms * 1e-3) / 1e9, 1)}


PROFILE = _Profile()


# ------------------------------------------------------------------------------------------------ dispatch policy (pcgcv2_amd/pathconfig.py)
from .pathconfig import PathConfig, FIELDS as _PATH_FIELDS

PATH = PathConfig.from_env()          # the process-wide default record: frozen; replaced as a whole, never mutated


def configure(cfg=None, **changes):
    """replace the default PathConfig: configure(FIELD=value, ...) or configure(a_record) -> the PREVIOUS record (hand it back to restore)"""
    global PATH
    prev = PATH
    PATH = (PATH if cfg is None else cfg).replace(**changes)
    return prev


class path:
    """with ops.path(ROWS_Q4=False): ...  — the default record replaced inside the block, the previous one restored on exit"""

    def __init__(self, **changes):
        self.changes = changes

    def __enter__(self):
        self.prev = configure(**self.changes)
        return PATH

    def __exit__(self, *exc):
        configure(self.prev)
        return False


class _OpsModule(type(_os)):
    """`ops.FIELD` reads the current record; `ops.FIELD = v` replaces the record (the spelling of the A/B tools and tests of rounds 1-5)"""

    def __getattr__(self, name):
        if name in _PATH_FIELDS:
            return getattr(self.__dict__['PATH'], name)
        raise AttributeError(f"module 'pcgcv2_amd.ops' has no attribute '{name}'")

    def __setattr__(self, name, value):
        if name in _PATH_FIELDS:
            self.__dict__['PATH'] = self.__dict__['PATH'].replace(**{name: value})
        else:
            super().__setattr__(name, value)


import sys as _sys
_sys.modules[__name__].__class__ = _OpsModule


# ------------------------------------------------------------------------------------------------ hash / coords
class HashTable:
    """Coordinate hash of one level (keys/vals device buffers + the stride its spatial blocking was built with)."""

    def __init__(self, coords, stride, keep_last=False):
        n = coords.shape[0]
        self.cap = int(lib().pcgc_hash_capacity(n))
        self.stride = int(stride)
        self.keys = torch.empty(self.cap, dtype=torch.int64, device=coords.device)
        self.vals = torch.empty(self.cap, dtype=torch.int32, device=coords.device)
        check(lib().pcgc_hash_clear(_p(self.keys), _p(self.vals), self.cap, _stream(self.keys)), 'hash_clear')
        check(lib().pcgc_hash_insert_policy(_p(_i32(coords)), n, self.stride, _p(self.keys), _p(self.vals), self.cap, int(keep_last),
                                            _stream(coords)), 'hash_insert')


def level_prepare_children(coords, stride):
    """hash table, k3 map, children level and the children level's k3 map of a level, in one library call
    -> (HashTable, nbr [27, n], children [8 n, 4], nbr_children [27, 8 n])"""
    n, dev = coords.shape[0], coords.device
    table = HashTable.__new__(HashTable)
    table.cap, table.stride = int(lib().pcgc_hash_capacity(n)), int(stride)
    table.keys = torch.empty(table.cap, dtype=torch.int64, device=dev)
    table.vals = torch.empty(table.cap, dtype=torch.int32, device=dev)
    ints = torch.empty(27 * n + 32 * n + 27 * 8 * n, dtype=torch.int32, device=dev)      # one allocation: nbr | children | nbr_children
    nbr, children, nbr_c = ints[:27 * n].view(27, n), ints[27 * n:59 * n].view(8 * n, 4), ints[59 * n:].view(27, 8 * n)
    check(lib().pcgc_level_prepare_children(_p(_i32(coords)), n, int(stride), _p(table.keys), _p(table.vals), table.cap, _p(nbr), _p(children),
                                            _p(nbr_c), _stream(coords)), 'level_prepare_children')
    return table, nbr, children, nbr_c


def check_coords(coords, what='coordinates'):
    """Raise PcgcError if any row is outside the range the coordinate key can hold (the hash kernels would skip it).
    -> descents of the (batch, z, y, x) key along the rows: 0 = sorted like sort_spare_tensor's output, ~n/2 = no order at all."""
    out2 = torch.empty(2, dtype=torch.int32, device=coords.device)
    check(lib().pcgc_coords_check_order(_p(_i32(coords)), coords.shape[0], _p(out2), _stream(coords)), 'coords_check_order')
    n_bad, descents = out2.tolist()
    if n_bad:
        raise PcgcError(f'{what}: {n_bad} of {coords.shape[0]} rows are outside the supported range '
                        '(0 <= x, y, z < 2^20, 0 <= batch < 16)')
    return int(descents)


def first_occurrence_mask(coords, table, want_rows=False):
    """-> keep uint8 [n] (and, if want_rows, first_row int32 [n]: the first row holding each row's coordinate)."""
    n = coords.shape[0]
    keep = torch.empty(n, dtype=torch.uint8, device=coords.device)
    first = torch.empty(n, dtype=torch.int32, device=coords.device) if want_rows else None
    check(lib().pcgc_hash_first_mask(_p(_i32(coords)), n, table.stride, _p(table.keys), _p(table.vals), table.cap, _p(keep),
                                     _p(first), _stream(coords)), 'hash_first_mask')
    return (keep, first) if want_rows else keep


def down_maps(fine, first_row, prefix, stride_fine, n_coarse):
    """-> (parent_of int32 [n_fine], down int32 [8, n_coarse]) of the k2s2 down-sampling, no hash probes."""
    n = fine.shape[0]
    parent_of = torch.empty(n, dtype=torch.int32, device=fine.device)
    down = torch.empty((8, n_coarse), dtype=torch.int32, device=fine.device)
    check(lib().pcgc_down_maps(_p(_i32(fine)), _p(first_row), _p(prefix), n, int(stride_fine), n_coarse, _p(parent_of), _p(down),
                               _stream(fine)), 'down_maps')
    return parent_of, down


def down_level(fine, stride_fine):
    """One strided pyramid level (MinkowskiConvolution k=2 s=2, coordinate side) in one library call, which reads the coarse
    count back in the middle (the one host synchronisation of a level) and finishes into upper-bound buffers
    -> (coarse int32 [n_coarse,4], parent_of int32 [n], down int32 [8,n_coarse]); coarse and down are views of buffers
    sized for n rows.  Canonical order: coarse rows in first-occurrence order of the quantised fine rows."""
    import ctypes
    fine = _i32(fine)
    n, dev = fine.shape[0], fine.device
    cap = int(lib().pcgc_hash_capacity(n))
    ws_bytes = int(lib().pcgc_scan_workspace_bytes(n))
    # two allocations per level (scratch, outputs) instead of eleven: this runs between two device synchronisations, where every
    # microsecond of interpreter time is GPU idle time
    a8 = lambda v: (v + 7) & ~7
    off, sizes = 0, []
    for nbytes in (16 * n, 8 * cap, 4 * cap, 4 * n, 4 * n, 8, n, ws_bytes):        # q | keys | vals | first_row | prefix | total | keep | scan ws
        sizes.append((off, nbytes)); off += a8(nbytes)
    scratch = torch.empty(off, dtype=torch.uint8, device=dev)
    part = lambda i, dt: scratch[sizes[i][0]:sizes[i][0] + sizes[i][1]].view(dt)
    q, keys, vals, first_row, prefix, total, keep, ws = (part(0, torch.int32), part(1, torch.int64), part(2, torch.int32), part(3, torch.int32),
                                                       part(4, torch.int32), part(5, torch.int32), part(6, torch.uint8), part(7, torch.uint8))
    outbuf = torch.empty(13 * n, dtype=torch.int32, device=dev)                     # coarse [n,4] | parent_of [n] | down [8n]
    coarse_ub, parent_of, down_ub = outbuf[:4 * n].view(n, 4), outbuf[4 * n:5 * n], outbuf[5 * n:]
    n_coarse = ctypes.c_int64(0)
    check(lib().pcgc_down_level(_p(fine), n, int(stride_fine), _p(q), _p(keys), _p(vals), cap, _p(keep), _p(first_row), _p(prefix),
                                _p(total), _p(ws), ws_bytes, _p(coarse_ub), _p(parent_of), _p(down_ub), ctypes.byref(n_coarse),
                                _stream(fine)), 'down_level')
    nc = int(n_coarse.value)
    return coarse_ub[:nc], parent_of, down_ub[:8 * nc].view(8, nc)


def pyramid(fine, stride_fine, levels):
    """`levels` strided levels below `fine` in one library call with ONE host synchronisation (pcgc_pyramid): every level is
    deduplicated straight from the input rows.  -> [(coarse, parent_of, down)] per level, the same tensors `levels` nested
    down_level() calls return (views of upper-bound buffers)."""
    import ctypes
    fine = _i32(fine)
    n, dev = fine.shape[0], fine.device
    nbytes = int(lib().pcgc_pyramid_scratch_bytes(n, levels))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    outbuf = torch.empty(levels * 13 * n, dtype=torch.int32, device=dev)            # per level: coarse [n,4] | parent_of [n] | down [8n]
    parts = [(outbuf[13 * n * l:13 * n * l + 4 * n].view(n, 4), outbuf[13 * n * l + 4 * n:13 * n * l + 5 * n], outbuf[13 * n * l + 5 * n:13 * n * (l + 1)])
             for l in range(levels)]
    ptrs = lambda k: (ctypes.c_void_p * levels)(*[p[k].data_ptr() for p in parts])
    counts = (ctypes.c_int64 * levels)()
    check(lib().pcgc_pyramid(_p(fine), n, int(stride_fine), int(levels), _p(scratch), nbytes, ptrs(0), ptrs(1), ptrs(2), counts, _stream(fine)),
          'pyramid')
    out, below = [], n
    for l, (coarse_ub, parent_ub, down_ub) in enumerate(parts):
        nc = int(counts[l])
        out.append((coarse_ub[:nc], parent_ub[:below], down_ub[:8 * nc].view(8, nc)))
        below = nc
    return out


def compact_index(mask, prefix, n_out):
    orig = torch.empty(n_out, dtype=torch.int32, device=mask.device)
    check(lib().pcgc_compact_index(_p(mask), _p(prefix), mask.shape[0], _p(orig), _stream(mask)), 'compact_index')
    return orig


def kmap_k3_children(parent_nbr):
    n_parent = parent_nbr.shape[1]
    nbr = torch.empty((27, 8 * n_parent), dtype=torch.int32, device=parent_nbr.device)
    check(lib().pcgc_kmap_k3_children(_p(parent_nbr), n_parent, _p(nbr), _stream(parent_nbr)), 'kmap_k3_children')
    return nbr


def kmap_k3_prune(cand_nbr, mask, prefix, orig):
    n_out = orig.shape[0]
    nbr = torch.empty((27, n_out), dtype=torch.int32, device=cand_nbr.device)
    check(lib().pcgc_kmap_k3_prune(_p(cand_nbr), cand_nbr.shape[1], _p(mask), _p(prefix), _p(orig), n_out, _p(nbr), _stream(cand_nbr)),
          'kmap_k3_prune')
    return nbr


def kmap_k3_prune_parent(parent_nbr, mask, prefix, orig):
    """k3 map of a pruned children level from the PARENT level's map (no [27][8 n_parent] candidate map)."""
    n_out = orig.shape[0]
    nbr = torch.empty((27, n_out), dtype=torch.int32, device=parent_nbr.device)
    check(lib().pcgc_kmap_k3_prune_parent(_p(parent_nbr), parent_nbr.shape[1], _p(mask), _p(prefix), _p(orig), n_out, _p(nbr),
                                          _stream(parent_nbr)), 'kmap_k3_prune_parent')
    return nbr


def kmap_k3_from_coarse(fine, stride_fine, parent_of, coarse_nbr, down):
    n = fine.shape[0]
    nbr = torch.empty((27, n), dtype=torch.int32, device=fine.device)
    check(lib().pcgc_kmap_k3_from_coarse(_p(_i32(fine)), n, int(stride_fine), _p(parent_of), _p(coarse_nbr), _p(down),
                                         coarse_nbr.shape[1], _p(nbr), _stream(fine)), 'kmap_k3_from_coarse')
    return nbr


def mask_scan(mask):
    """-> (prefix int32 [n], total int32 [1] on device)."""
    n = mask.shape[0]
    prefix = torch.empty(n, dtype=torch.int32, device=mask.device)
    total = torch.empty(1, dtype=torch.int32, device=mask.device)
    ws_bytes = int(lib().pcgc_scan_workspace_bytes(n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=mask.device)
    check(lib().pcgc_mask_scan(_p(_dev(mask, torch.uint8, 'mask')), n, _p(prefix), _p(total), _p(ws), ws_bytes, _stream(mask)),
          'mask_scan')
    return prefix, total


def compact_coords(coords, mask, prefix, n_out):
    out = torch.empty((n_out, 4), dtype=torch.int32, device=coords.device)
    check(lib().pcgc_compact_coords(_p(_i32(coords)), _p(mask), _p(prefix), coords.shape[0], _p(out), _stream(coords)),
          'compact_coords')
    return out


def compact_feats(feats, mask, prefix, n_out):
    _f32(feats)
    C = feats.shape[1]
    out = torch.empty((n_out, C), dtype=torch.float32, device=feats.device)
    check(lib().pcgc_compact_feats(_p(feats), C, _ld(feats), _p(mask), _p(prefix), feats.shape[0], _p(out), _stream(feats)),
          'compact_feats')
    return out


def coords_quantize(coords, stride_out):
    out = torch.empty_like(coords)
    check(lib().pcgc_coords_quantize(_p(_i32(coords)), coords.shape[0], int(stride_out), _p(out), _stream(coords)), 'coords_quantize')
    return out


def coords_children(coords, stride_in):
    out = torch.empty((8 * coords.shape[0], 4), dtype=torch.int32, device=coords.device)
    check(lib().pcgc_coords_children(_p(_i32(coords)), coords.shape[0], int(stride_in), _p(out), _stream(coords)), 'coords_children')
    return out


def coords_scale(coords, factor):
    out = torch.empty_like(coords)
    check(lib().pcgc_coords_scale(_p(_i32(coords)), coords.shape[0], float(np.float32(factor)), _p(out), _stream(coords)),
          'coords_scale')
    return out


def kmap_k3(coords, stride, table):
    n = coords.shape[0]
    nbr = torch.empty((27, n), dtype=torch.int32, device=coords.device)
    check(lib().pcgc_kmap_k3(_p(_i32(coords)), n, int(stride), _p(table.keys), _p(table.vals), table.cap, _p(nbr), _stream(coords)),
          'kmap_k3')
    return nbr


# ------------------------------------------------------------------------------------------------ conv family
def set_conv_impl(impl):
    """family of pcgc_conv_gather: -1 auto, 0 = direct-load VALU kernel, 2 = LDS-DMA + MFMA, 6 = row-split (bit-identical results)."""
    check(lib().pcgc_set_conv_impl(int(impl)), 'set_conv_impl')


def conv_gather(nbr, x, W, bias, out=None, residual=None, relu=False, n_out=None):
    """out (view, may be a column slice) = relu?( fmaf-chain(nbr, x, W) + bias (+ residual) )."""
    _f32(x, 'x'); _f32(W, 'W')
    if W.dim() == 2:
        K, (Cin, Cout) = 1, W.shape
    else:
        K, Cin, Cout = W.shape
    if x.shape[1] != Cin:
        raise PcgcError(f'conv_gather: input has {x.shape[1]} channels, kernel expects {Cin}')
    if nbr is None:
        n_out = x.shape[0]
    else:
        n_out = nbr.shape[1]
        if nbr.shape[0] != K:
            raise PcgcError('conv_gather: kernel map / kernel volume mismatch')
    if out is None:
        out = torch.empty((n_out, Cout), dtype=torch.float32, device=x.device)
    res_p, res_ld = (None, 0) if residual is None else (_p(_f32(residual)), _ld(residual))
    prof = K == 27 and PROFILE.want(('conv', Cin, Cout, n_out))
    if prof:
        mfma = Cin in (16, 32, 64) and Cout in (16, 32, 64) and n_out >= 8192
        e0, e1 = PROFILE.bracket(('conv', Cin, Cout, n_out), f'k3 gather conv Cin={Cin} Cout={Cout} (k_conv_gather_mfma*/dma, per-row gather)', n_out,
                                 lambda P, a=Cin, b=Cout, n=n_out: P * a * 4 + P * 8 + n * b * 4,
                                 lambda P, a=Cin, b=Cout: 2 * P * a * b,
                                 compulsory=x.shape[0] * Cin * 4 + 27 * n_out * 4 + n_out * Cout * 4,
                                 mfma_issued=(2 * 27 * ((n_out + 15) // 16 * 16) * Cin * Cout) if mfma else None)
        e0.record()
    check(lib().pcgc_conv_gather(_p(nbr), K, n_out, _p(x), x.shape[0], Cin, _ld(x), 0, _p(W), _p(bias), res_p, res_ld, 0, int(relu),
                                 _p(out), Cout, _ld(out), 0, _stream(nbr)), 'conv_gather')
    if prof:
        e1.record()
    elif PROFILE.counting and K == 27:
        PROFILE.count(nbr)
    return out




def conv_gather_unit(nbr, W, bias, relu=False):
    """conv_gather for the all-ones single-channel input (the occupancy indicator every coded cloud starts from):
    out = relu?(sum of W[k][0][:] over the present offsets k, ascending, + bias)."""
    _f32(W, 'W')
    K, Cin, Cout = W.shape
    if Cin != 1 or nbr.shape[0] != K:
        raise PcgcError('conv_gather_unit: kernel [K, 1, Cout] and a [K, n] map expected')
    n_out = nbr.shape[1]
    out = torch.empty((n_out, Cout), dtype=torch.float32, device=W.device)
    key = ('conv', 1, Cout, n_out)
    prof = K == 27 and PROFILE.want(key)
    if prof:
        e0, e1 = PROFILE.bracket(key, f'k3 conv 1->{Cout} on the unit input (k_conv_unit: kernel map only, no feature gather)', n_out,
                                 lambda P, b=Cout, n=n_out: P * 4 + P * 8 + n * b * 4, lambda P, b=Cout: 2 * P * b,
                                 compulsory=27 * n_out * 4 + n_out * Cout * 4)
        e0.record()
    check(lib().pcgc_conv_gather_unit(_p(nbr), K, n_out, _p(W), _p(bias), int(relu), _p(out), Cout, Cout, _stream(nbr)), 'conv_gather_unit')
    if prof:
        e1.record()
    elif PROFILE.counting and K == 27:
        PROFILE.count(nbr)
    return out


def conv_unit_from_coarse(fine, stride_fine, parent_of, coarse_nbr, down, W, bias, relu=False):
    """conv_gather_unit on a pyramid level whose own k3 map was never built: presence from the parent level's map + the down map
    (pcgc_conv_unit_from_coarse); same sums, same order."""
    _f32(W, 'W')
    K, Cin, Cout = W.shape
    if Cin != 1 or K != 27:
        raise PcgcError('conv_unit_from_coarse: kernel [27, 1, Cout] expected')
    n = fine.shape[0]
    out = torch.empty((n, Cout), dtype=torch.float32, device=W.device)
    key = ('conv', 1, Cout, n)
    prof = PROFILE.want(key)
    if prof:
        e0, e1 = PROFILE.bracket(key, f'k3 conv 1->{Cout} on the unit input (k_conv_unit_coarse: presence from the parent level\'s map, no map of its own)', n,
                                 lambda P, b=Cout, n=n: P * 4 + P * 8 + n * b * 4, lambda P, b=Cout: 2 * P * b,
                                 compulsory=n * 16 + n * 4 + n * Cout * 4)
        e0.record()
    check(lib().pcgc_conv_unit_from_coarse(_p(_i32(fine)), n, int(stride_fine), _p(parent_of), _p(coarse_nbr), _p(down), coarse_nbr.shape[1],
                                           _p(W), _p(bias), int(relu), _p(out), Cout, Cout, _stream(fine)), 'conv_unit_from_coarse')
    if prof:
        e1.record()
    return out


def set_up2_impl(mfma):
    """generative transpose conv kernel for 64->32 / 32->16: 2 fp32 MFMA with LDS-resident fragments (default), 1 fragments from L2, 0 VALU."""
    check(lib().pcgc_set_up2_impl(int(mfma)), 'set_up2_impl')




def _irn_pass_formulas(n, C, map_bytes, names):
    """[(pass, name, gathered_bytes(P), flops(P), compulsory_bytes)] of the two fused InceptionResNet passes on n rows."""
    Q = C // 4
    return ((1, names[0], lambda P: (P * C * 4 + P * 8 + n * Q * 4) + n * (C + Q) * 4, lambda P: 2 * P * C * Q + 2 * n * C * Q,
             n * C * 4 + map_bytes + n * 2 * Q * 4),
            (2, names[1], lambda P: (P * Q * 4 + P * 8 + n * 2 * Q * 4) + (P * Q * 4 + P * 8 + n * Q * 4) + n * 3 * Q * 4,
             lambda P: 2 * P * Q * 2 * Q + 2 * P * Q * Q + 2 * n * Q * 2 * Q,
             n * 2 * Q * 4 + map_bytes + n * C * 4 + n * C * 4))


def irn_block(nbr, x, params):
    """Fused InceptionResNet block; params = [conv0_0.kernel, .bias, conv0_1.kernel, .bias, conv1_0..., conv1_1..., conv1_2...]."""
    import ctypes
    _f32(x, 'x')
    n, C = x.shape
    t = torch.empty((n, C // 2), dtype=torch.float32, device=x.device)
    out = torch.empty((n, C), dtype=torch.float32, device=x.device)
    arr = (ctypes.c_void_p * 10)(*[p.data_ptr() for p in params])
    if PROFILE.counting:
        PROFILE.count(nbr)
    name_a, name_b = (f'k_irn_a<{C}, 16>', f'k_irn_b<{C}, 16>') if C == 64 else (f'k_irn_a_split<{C}>', f'k_irn_b_split<{C}>')
    if not (PROFILE.want((name_a, n)) or PROFILE.want((name_b, n))):
        check(lib().pcgc_irn_block(_p(nbr), n, _p(x), C, _ld(x), arr, _p(t), _p(out), C, _stream(nbr)), 'irn_block')
        return out
    Q = C // 4
    passes = _irn_pass_formulas(n, C, 27 * n * 4, (name_a, name_b))
    for ps, name, bf, ff, comp in passes:
        prof = PROFILE.want((name, n))
        if prof:
            e0, e1 = PROFILE.bracket((name, n), name + ' (per-row gather, VALU)', n, bf, ff, compulsory=comp)
            e0.record()
        check(lib().pcgc_irn_pass(_p(nbr), n, _p(x), C, _ld(x), arr, _p(t), _p(out), C, ps, _stream(nbr)), 'irn_pass')
        if prof:
            e1.record()
    return out




def irn_block_rows64(nbr, x, params, tables):
    """C = 64 InceptionResNet on a plain level through its own k3 map (k_rows_irn_a64, k_rows_irn_b64; tables = child_irn_tables(params));
    bit-identical to irn_block."""
    _f32(x, 'x')
    n = x.shape[0]
    ta, tb = tables
    t = torch.empty((n, 32), dtype=torch.float32, device=x.device)
    out = torch.empty((n, 64), dtype=torch.float32, device=x.device)
    P = [p.data_ptr() for p in params]
    s = _stream(x)
    if PROFILE.counting:
        PROFILE.count(nbr)
    tiles = (n + 15) // 16
    forms = _irn_pass_formulas(n, 64, 27 * n * 4, ('k_rows_irn_a64', 'k_rows_irn_b64'))
    per_tile = (27 * 16 + 16, 27 * 12 + 8)                   # MFMA instructions per 16-row tile (pass B incl. the conv1_2 products)
    calls = (lambda: lib().pcgc_irn_rows_pass(_p(nbr), n, 64, 1, _p(x), _ld(x), _p(ta), ta.numel() * 4, P[1], P[5], None, None, 0, _p(t), 32, s),
             lambda: lib().pcgc_irn_rows_pass(_p(nbr), n, 64, 2, _p(t), 32, _p(tb), tb.numel() * 4, P[3], P[7], P[9], _p(x), _ld(x), _p(out), 64, s))
    for (ps, name, bf, ff, comp), call, mf in zip(forms, calls, per_tile):
        prof = PROFILE.want((name, n))
        if prof:
            e0, e1 = PROFILE.bracket((name, n), name + ' (fused InceptionResNet pass on a plain level, LDS-resident table + per-wave row ring, fp32 MFMA)', n, bf, ff,
                                     compulsory=comp, mfma_issued=tiles * mf * 2048)
            e0.record()
        check(call(), 'irn_rows_pass')
        if prof:
            e1.record()
    return out


def _rows_irn32_index():
    """Gather indices of the two pass tables of the plain-level C = 32 InceptionResNet (csrc/rows_irn.hip: RowsPassA32 / RowsPassB32)."""
    C, Q = 32, 8
    shapes = [(27, C, Q), (27, Q, 2 * Q), (C, Q), (27, Q, Q), (Q, 2 * Q)]
    off, Ws = 0, []
    for shp in shapes:
        n = int(np.prod(shp))
        Ws.append(np.arange(off, off + n, dtype=np.int64).reshape(shp))
        off += n
    W00, W01, W10, W11, W12 = Ws
    # pass A: one fragment per offset and 16-channel block: columns 0-7 conv0_0, columns 8-15 conv1_0 (centre offset only)
    fa = [_fragment([W00[k][:, co] for co in range(Q)] + [(W10[:, co] if k == 13 else None) for co in range(Q)], 2) for k in range(27)]
    # pass B: t rows are 16 wide: channels 0-7 (K-steps 0, 1) feed conv0_1, channels 8-15 (K-steps 2, 3) conv1_1 (columns 8-15 zero)
    pad = lambda w: np.concatenate([np.full(Q, -1, np.int64), w])
    fb = [_fragment([W01[k][:, co] for co in range(16)], 1, KS=2, k0=0) for k in range(27)]
    fb += [_fragment([pad(W11[k][:, co]) for co in range(Q)] + [None] * Q, 1, KS=2, k0=2) for k in range(27)]
    fb.append(_fragment([W12[:, co] for co in range(16)], 1, KS=2, k0=0))
    return np.concatenate([f.reshape(-1) for f in fa]), np.concatenate([f.reshape(-1) for f in fb])


def rows_irn32_tables(params):
    """(table A 54 KB, table B 27.5 KB) of the plain-level C = 32 InceptionResNet passes; params as in irn_block."""
    W00, b00, W01, b01, W10, b10, W11, b11, W12, b12 = params
    key = ('rows_irn32', W00.device)
    if key not in _TABLE_INDEX:
        ia, ib = _rows_irn32_index()
        _TABLE_INDEX[key] = (torch.from_numpy(ia).to(W00.device), torch.from_numpy(ib).to(W00.device))
    flat = torch.cat([w.detach().reshape(-1) for w in (W00, W01, W10, W11, W12)])
    ia, ib = _TABLE_INDEX[key]
    return _gather_table(ia, flat), _gather_table(ib, flat)




def irn_block_rows32(nbr, x, params, tables):
    """C = 32 InceptionResNet on a plain level through its own k3 map (k_rows_irn_a32 / _b32); bit-identical to irn_block."""
    _f32(x, 'x')
    n = x.shape[0]
    ta, tb = tables
    t = torch.empty((n, 16), dtype=torch.float32, device=x.device)
    out = torch.empty((n, 32), dtype=torch.float32, device=x.device)
    P = [p.data_ptr() for p in params]
    s = _stream(x)
    if PROFILE.counting:
        PROFILE.count(nbr)
    tiles = (n + 15) // 16
    forms = _irn_pass_formulas(n, 32, 27 * n * 4, ('k_rows_irn_a32', 'k_rows_irn_b32'))
    per_tile = (27 * 8, 27 * 4 + 2)                          # MFMA instructions per 16-row tile (pass A: 8 K-steps per offset; pass B: 2 + 2, + conv1_2)
    calls = (lambda: lib().pcgc_irn_rows_pass(_p(nbr), n, 32, 1, _p(x), _ld(x), _p(ta), ta.numel() * 4, P[1], P[5], None, None, 0, _p(t), 16, s),
             lambda: lib().pcgc_irn_rows_pass(_p(nbr), n, 32, 2, _p(t), 16, _p(tb), tb.numel() * 4, P[3], P[7], P[9], _p(x), _ld(x), _p(out), 32, s))
    for (ps, name, bf, ff, comp), call, mf in zip(forms, calls, per_tile):
        prof = PROFILE.want((name, n))
        if prof:
            e0, e1 = PROFILE.bracket((name, n), name + ' (fused InceptionResNet pass on a plain level, LDS-resident table + per-wave row ring, packed-N fp32 MFMA)', n, bf, ff,
                                     compulsory=comp, mfma_issued=tiles * mf * 2048)
            e0.record()
        check(call(), 'irn_rows_pass')
        if prof:
            e1.record()
    return out


def rows32_pass(nbr, x, params, tables, ps, t=None):
    """one pass of irn_block_rows32 on its own (A/B tools): ps = 1 -> t [n, 16]; ps = 2 (t given) -> out [n, 32]"""
    n = x.shape[0]
    ta, tb = tables
    P = [p.data_ptr() for p in params]
    s = _stream(x)
    if ps == 1:
        t = torch.empty((n, 16), dtype=torch.float32, device=x.device)
        check(lib().pcgc_irn_rows_pass(_p(nbr), n, 32, 1, _p(x), _ld(x), _p(ta), ta.numel() * 4, P[1], P[5], None, None, 0, _p(t), 16, s), 'irn_rows_pass')
        return t
    out = torch.empty((n, 32), dtype=torch.float32, device=x.device)
    check(lib().pcgc_irn_rows_pass(_p(nbr), n, 32, 2, _p(t), 16, _p(tb), tb.numel() * 4, P[3], P[7], P[9], _p(x), _ld(x), _p(out), 32, s), 'irn_rows_pass')
    return out


def rows_q4_tables(params):
    """(table A 28 672 B, table B 21 248 B) of the quad-block C = 32 InceptionResNet passes on plain levels (csrc/rows_q4.hip); params as in
    irn_block.  Every fragment is [co 4][16 floats]: a lane's operands of one group are its output channel's 16 values, contiguous.
      A: [k = 0..26][channel half h][output group g] = W00[k][16 h + j][4 g + co], then [h][g] = W10[16 h + j][4 g + co]
      B: [k][X0, X1, Y]: X_p[co][8 half + j] = W01[k][j][4 (2 p + half) + co], Y[co][8 half + j] = W11[k][j][4 half + co];
         then conv1_2's two: [p][co][8 half + j] = W12[j][4 (2 p + half) + co]."""
    W00, b00, W01, b01, W10, b10, W11, b11, W12, b12 = params
    C = W00.shape[1]
    if C != 32:
        raise PcgcError('rows_q4_tables: C = 32')
    a0 = W00.detach().reshape(27, 2, 16, 2, 4).permute(0, 1, 3, 4, 2).reshape(-1)                 # [k][h][g][co][j]
    a1 = W10.detach().reshape(2, 16, 2, 4).permute(0, 2, 3, 1).reshape(-1)                          # [h][g][co][j]
    x01 = W01.detach().reshape(27, 8, 2, 2, 4).permute(0, 2, 4, 3, 1).reshape(27, 2, 64)            # [k][p][co][half][j]
    y11 = W11.detach().reshape(27, 8, 2, 4).permute(0, 3, 2, 1).reshape(27, 1, 64)                  # [k][co][half][j]
    w12 = W12.detach().reshape(8, 2, 2, 4).permute(1, 3, 2, 0).reshape(-1)                          # [p][co][half][j]
    return torch.cat([a0, a1]).contiguous(), torch.cat([torch.cat([x01, y11], 1).reshape(-1), w12]).contiguous()




def rows_q4_pass(nbr, x, params, tables, ps, t=None):
    """one pass of irn_block_rows32_q4 on its own (A/B tools)"""
    n = x.shape[0]
    ta, tb = tables
    P = [p.data_ptr() for p in params]
    s = _stream(x)
    if ps == 1:
        t = torch.empty((n, 16), dtype=torch.float32, device=x.device)
        check(lib().pcgc_irn_rows_q4_pass(_p(nbr), n, 32, 1, _p(x), _ld(x), _p(ta), ta.numel() * 4, P[1], P[5], None, None, 0, _p(t), 16, s), 'irn_rows_q4_pass')
        return t
    out = torch.empty((n, 32), dtype=torch.float32, device=x.device)
    check(lib().pcgc_irn_rows_q4_pass(_p(nbr), n, 32, 2, _p(t), 16, _p(tb), tb.numel() * 4, P[3], P[7], P[9], _p(x), _ld(x), _p(out), 32, s), 'irn_rows_q4_pass')
    return out


def irn_block_rows32_q4(nbr, x, params, tables):
    """C = 32 InceptionResNet on a plain level through its own k3 map, quad-block form (k_rows_q4_a32 / _b32); bit-identical to irn_block."""
    _f32(x, 'x')
    n = x.shape[0]
    ta, tb = tables
    t = torch.empty((n, 16), dtype=torch.float32, device=x.device)
    out = torch.empty((n, 32), dtype=torch.float32, device=x.device)
    P = [p.data_ptr() for p in params]
    s = _stream(x)
    if PROFILE.counting:
        PROFILE.count(nbr)
    tiles = (n + 63) // 64
    forms = _irn_pass_formulas(n, 32, 27 * n * 4, ('k_rows_q4_a32', 'k_rows_q4_b32'))
    per_tile = (112 * 16, 81 * 16 + 32)                      # 4x4x1 instructions (512 flops each) per 64-row M tile
    calls = (lambda: lib().pcgc_irn_rows_q4_pass(_p(nbr), n, 32, 1, _p(x), _ld(x), _p(ta), ta.numel() * 4, P[1], P[5], None, None, 0, _p(t), 16, s),
             lambda: lib().pcgc_irn_rows_q4_pass(_p(nbr), n, 32, 2, _p(t), 16, _p(tb), tb.numel() * 4, P[3], P[7], P[9], _p(x), _ld(x), _p(out), 32, s))
    for (ps, name, bf, ff, comp), call, mf in zip(forms, calls, per_tile):
        prof = PROFILE.want((name, n))
        if prof:
            e0, e1 = PROFILE.bracket((name, n), name + ' (fused InceptionResNet pass on a plain level, quad-block 4x4x1 fp32 MFMA, lane = row)', n, bf, ff,
                                     compulsory=comp, mfma_issued=tiles * mf * 512)
            e0.record()
        check(call(), 'irn_rows_q4_pass')
        if prof:
            e1.record()
    return out


def set_rows_q4_variant(v):
    check(lib().pcgc_set_rows_q4_variant(int(v)), 'set_rows_q4_variant')


# ------------------------------------------------------------------------------------------------ children-level convs


def _halo_cells():
    """The 64 cells of a parent's 4x4x4 halo in ascending (cz, cy, cx) order -> list of (kp, j', reach) where kp = index of the
    neighbour parent in the PARENT level's k3 map, j' = which of its children the cell is, and reach = [(j, k)]: the children j
    of the centre parent whose 3x3x3 window contains the cell, with the kernel offset k they see it through."""
    P1 = (0, 1, 1, 2)                  # floor(c / 2) + 1 for c = -1, 0, 1, 2
    B = (1, 0, 1, 0)                   # c & 1
    cells = []
    for cz in range(4):
        for cy in range(4):
            for cx in range(4):
                kp = P1[cz] * 9 + P1[cy] * 3 + P1[cx]
                jc = B[cx] + 2 * B[cy] + 4 * B[cz]
                reach = []
                for j in range(8):
                    kx, ky, kz = cx - (j & 1), cy - ((j >> 1) & 1), cz - (j >> 2)          # = (c - j) + 1 per axis
                    if 0 <= kx <= 2 and 0 <= ky <= 2 and 0 <= kz <= 2:
                        reach.append((j, kz * 9 + ky * 3 + kx))
                cells.append((kp, jc, reach))
    assert sum(len(r) for _, _, r in cells) == 216
    return cells


def child_conv_table(W):
    """B-fragment table of pcgc_conv_child for a plain k3 conv `kernel` [27, Cin, Cout]:
    table[k][n][cb][lane][jj] = W[k][16 cb + 4 jj + (lane >> 4)][16 n + (lane & 15)]   (one lane-linear 1 KB fragment per
    (offset, column tile, 16-channel block): a lane's four K-step values are contiguous -> one conflict-free ds_read_b128)."""
    K, Cin, Cout = W.shape                                     # (K = 27; 8 for the k2 s2 down convs of conv_down_rows)
    NB, NT = Cin // 16, Cout // 16
    return W.detach().reshape(K, NB, 4, 4, NT, 16).permute(0, 4, 1, 3, 5, 2).contiguous().reshape(-1)       # [k][n][cb][mq][mi][jj]


def _fragment(col_weights, NB, KS=4, k0=0, half=False):
    """One B fragment per 16-channel block from `col_weights`: a list of 16 (8 if half) columns, each None (zero column) or an
    array [Cin_rows] giving that column's weight per input channel.  Layout [cb][lane = mq*16 + mi (mq*8 + mi if half)][jj]:
    value = column mi at input channel 16 cb + 4 (k0 + jj) + mq.  Entries nothing maps to are -1 (the builders below run on
    INDEX-valued weights: each "weight" is its own position in the flat parameter vector, so the result is a gather index)."""
    ncol = 8 if half else 16
    out = np.full((NB, 4, ncol, KS), -1, np.int64)
    for mi, w in enumerate(col_weights):
        if w is None:
            continue
        for cb in range(NB):
            for jj in range(KS):
                for mq in range(4):
                    ch = 16 * cb + 4 * (k0 + jj) + mq
                    if ch < len(w):
                        out[cb, mq, mi, jj] = w[ch]
    return out.reshape(NB, -1)


_TABLE_INDEX = {}          # (kind, C, device) -> int64 gather index (-1 = zero) into the flat parameter vector


def _gather_table(index, flat):
    return torch.where(index >= 0, flat[index.clamp(min=0)], torch.zeros((), dtype=flat.dtype, device=flat.device)).contiguous()


def _cls_index(C):
    """gather index of the compact classification-head table (csrc/child_kernels.h: ClsHead): 125 rows k' = 25 (kz + 1) + 5 (ky + 1) + (kx + 1),
    kz, ky, kx in -1 .. 3; a row with all three in 0 .. 2 is kernel[9 kz + 3 ky + kx] as [cb][mq][jj] -> channel 16 cb + 4 jj + mq, every other
    row is zero; rows are (C / 16) * 64 + 16 bytes apart, the table is padded to a multiple of 1 KB."""
    nb = C // 16
    rowf = nb * 16 + 4
    idx = np.full(((125 * rowf * 4 + 1023) // 1024 * 256,), -1, np.int64)
    for kz in range(3):
        for ky in range(3):
            for kx in range(3):
                r = 25 * (kz + 1) + 5 * (ky + 1) + (kx + 1)
                k = 9 * kz + 3 * ky + kx
                for cb in range(nb):
                    for mq in range(4):
                        for jj in range(4):
                            idx[r * rowf + cb * 16 + mq * 4 + jj] = k * C + 16 * cb + 4 * jj + mq
    return idx


def child_cls_table(W):
    """Table of the classification head (k3 conv C -> 1) for pcgc_conv_child: the kernel in a zero-padded 5 x 5 x 5 offset space (one
    ds_read_b128 per lane fetches the weights through which ITS child sees a cell, or zeros).  Built by one device gather through a
    cached index."""
    C = W.shape[1]
    key = ('cls', C, W.device)
    if key not in _TABLE_INDEX:
        _TABLE_INDEX[key] = torch.from_numpy(_cls_index(C)).to(W.device)
    return _gather_table(_TABLE_INDEX[key], W.detach().reshape(-1))


def _q4_cls_index(C):
    """gather index of the quad-block classification head's table: one fragment [s 4][ci C] per (cell, z half h) group in the kernel's
    schedule order (cells ascending; h = 0 if the cell's z plane is reached by the children with z bit 0, then h = 1): column s =
    kernel[k(cell, 4 h + s)][:, 0] where child 4 h + s reaches the cell, zero elsewhere."""
    W = np.arange(27 * C, dtype=np.int64).reshape(27, C)
    frags = []
    for c, (kp, jc, reach) in enumerate(_halo_cells()):
        cz = c >> 4
        kof = dict(reach)
        for h in range(2):
            if not 0 <= cz - h <= 2:
                continue
            f = np.full((4, C), -1, np.int64)
            for sidx in range(4):
                if 4 * h + sidx in kof:
                    f[sidx] = W[kof[4 * h + sidx]]
            frags.append(f.reshape(-1))
    assert len(frags) == 96
    return np.concatenate(frags)


def child_q4_cls_table(W):
    """Table of pcgc_cls_child_q4 (k3 conv 16 -> 1 on a children level, quad-block form)."""
    C = W.shape[1]
    key = ('q4cls', C, W.device)
    if key not in _TABLE_INDEX:
        _TABLE_INDEX[key] = torch.from_numpy(_q4_cls_index(C)).to(W.device)
    return _gather_table(_TABLE_INDEX[key], W.detach().reshape(-1))


def cls_child_q4(parent_nbr, x, table, bias):
    """Classification head k3 16 -> 1 on the children level of `parent_nbr`'s level in quad-block form (csrc/child_q4.h): -> [8 n_parent, 1]."""
    _f32(x, 'x')
    n_p = parent_nbr.shape[1]
    if x.shape[0] != 8 * n_p:
        raise PcgcError('cls_child_q4: feature rows must be 8 x the parent level')
    Cin = x.shape[1]
    n = 8 * n_p
    out = torch.empty((n, 1), dtype=torch.float32, device=x.device)
    key = ('child_conv', Cin, 1, n)
    prof = PROFILE.want(key)
    if prof:
        e0, e1 = PROFILE.bracket(key, f'k_child_q4<1, 8, 2> (k3 {Cin}->1 on a children level, parent-map halo gather + quad-block 4x4x1 fp32 MFMA)', n,
                                 lambda P, a=Cin: P * a * 4 + P * 8 + n * 4, lambda P, a=Cin: 2 * P * a,
                                 compulsory=n * Cin * 4 + 27 * n_p * 4 + n * 4, mfma_issued=((n_p + 63) // 64) * 96 * 16 * 512)
        e0.record()
    check(lib().pcgc_cls_child_q4(_p(parent_nbr), n_p, _p(x), Cin, _ld(x), _p(table), table.numel() * 4, _p(bias), _p(out), _stream(x)), 'cls_child_q4')
    if prof:
        e1.record()
    elif PROFILE.counting:
        PROFILE.count_children(parent_nbr)
    return out


def _irn_index(C):
    """Gather indices (into cat(W00, W01, W10, W11, W12) flattened) of the two pass tables; the fragment order is the one
    csrc/child_kernels.h's PassA / PassB variants index (frag())."""
    Q, NB = C // 4, C // 16
    shapes = [(27, C, Q), (27, Q, 2 * Q), (C, Q), (27, Q, Q), (Q, 2 * Q)]
    off, Ws = 0, []
    for shp in shapes:
        n = int(np.prod(shp))
        Ws.append(np.arange(off, off + n, dtype=np.int64).reshape(shp))
        off += n
    W00, W01, W10, W11, W12 = Ws
    cpt = 16 // Q                                             # children per Q-wide tile: 4 (C=16) or 2 (C=32)

    def packed(Wk3, width, kz, ky, cy, cx):
        """columns (child-in-tile, channel) of a tile whose children differ in x only (ky given) or in (y, x) (ky None):
        -> list of 16 columns for the cell at (cy, cx) seen through z-offset kz"""
        cols = []
        nchild = 16 // width
        for sub in range(nchild):
            if ky is None:                                     # tile = z half: sub = jy*2 + jx
                jy, jx = sub >> 1, sub & 1
                kyy = cy - jy
            else:                                              # tile = (z, y) quarter: sub = jx
                jx, kyy = sub, ky
            kxx = cx - jx
            ok = 0 <= kyy <= 2 and 0 <= kxx <= 2
            for co in range(width):
                cols.append(Wk3[kz * 9 + kyy * 3 + kxx][:, co] if ok else None)
        return cols

    # ---- pass A: conv0_0 (k3 C -> Q) fragments, then conv1_0 (k1) fragments for the cell that is the child itself
    fa = []
    if cpt == 1:                                               # C = 64: one child per tile = the offset's own slice; pass B stays on the per-row kernels
        for k in range(27):
            fa.append(_fragment([W00[k][:, co] for co in range(16)], NB))
        fa.append(_fragment([W10[:, co] for co in range(16)], NB))
        # pass B: conv0_1 (k3 16 -> 32) fragments (k, n), conv1_1 (k3 16 -> 16) fragments k, conv1_2 (k1 16 -> 32) fragments n: one 16-channel block each
        fb = []
        for k in range(27):
            for n2 in range(2):
                fb.append(_fragment([W01[k][:, 16 * n2 + co] for co in range(16)], 1))
        for k in range(27):
            fb.append(_fragment([W11[k][:, co] for co in range(16)], 1))
        for n2 in range(2):
            fb.append(_fragment([W12[:, 16 * n2 + co] for co in range(16)], 1))
        return np.concatenate([f.reshape(-1) for f in fa]), np.concatenate([f.reshape(-1) for f in fb])
    if cpt == 4:
        for kz in range(3):
            for cy in range(4):
                for cx in range(4):
                    fa.append(_fragment(packed(W00, Q, kz, None, cy, cx), NB))
        for vy in range(2):
            for vx in range(2):
                fa.append(_fragment([W10[:, co] if sub == vy * 2 + vx else None for sub in range(4) for co in range(Q)], NB))
    else:
        for kz in range(3):
            for ky in range(3):
                for cx in range(4):
                    fa.append(_fragment(packed(W00, Q, kz, ky, None, cx), NB))
        for vx in range(2):
            fa.append(_fragment([W10[:, co] if sub == vx else None for sub in range(2) for co in range(Q)], NB))
    table_a = np.concatenate([f.reshape(-1) for f in fa])
    # ---- pass B: input rows are t (2Q wide): conv0_1 reads channels [0, Q) = K-steps [0, KS), conv1_1 channels [Q, 2Q) = [KS, 2KS)
    KS = Q // 4
    H = 2 * Q
    fb = []
    pad = lambda w, lo: np.concatenate([np.full(lo, -1, np.int64), w])          # place a Q-vector at channel offset lo of the 2Q row
    if H == 16:                                                # C = 32: conv0_1 tile = one child, 16 columns
        for k in range(27):
            fb.append(_fragment([W01[k][:, co] for co in range(16)], 1, KS=KS, k0=0))
    else:                                                      # C = 16: conv0_1 tile = (z, y) quarter, (jx, 8 columns)
        for kz in range(3):
            for ky in range(3):
                for cx in range(4):
                    fb.append(_fragment(packed(W01, H, kz, ky, None, cx), 1, KS=KS, k0=0))
    W11p = np.stack([np.stack([pad(W11[k][:, co], Q) for co in range(Q)], 1) for k in range(27)])      # [27][2Q][Q]: rows Q.. hold W11
    if cpt == 2:                                               # C = 32: conv1_1 tile = (z, y) quarter, (jx, 8 columns)
        for kz in range(3):
            for ky in range(3):
                for cx in range(4):
                    fb.append(_fragment(packed(W11p, Q, kz, ky, None, cx), 1, KS=KS, k0=KS))
    else:                                                      # C = 16: conv1_1 tile = z half, (jy jx, 4 columns)
        for kz in range(3):
            for cy in range(4):
                for cx in range(4):
                    fb.append(_fragment(packed(W11p, Q, kz, None, cy, cx), 1, KS=KS, k0=KS))
    fb.append(_fragment([W12[:, co] if co < H else None for co in range(16)], 1, KS=KS, k0=0))     # conv1_2 (k1 Q -> 2Q)
    table_b = np.concatenate([f.reshape(-1) for f in fb])
    return table_a, table_b


def child_irn_tables(params):
    """(table A, table B) of the parent-map InceptionResNet passes (C = 16 or 32); params as in irn_block.  Two device gathers
    through cached indices: cheap enough to redo whenever a checkpoint is loaded (R-D sweeps load one per rate)."""
    W00, b00, W01, b01, W10, b10, W11, b11, W12, b12 = params
    C = W00.shape[1]
    key = ('irn', C, W00.device)
    if key not in _TABLE_INDEX:
        ia, ib = _irn_index(C)
        _TABLE_INDEX[key] = (torch.from_numpy(ia).to(W00.device), None if ib is None else torch.from_numpy(ib).to(W00.device))
    flat = torch.cat([w.detach().reshape(-1) for w in (W00, W01, W10, W11, W12)])
    ia, ib = _TABLE_INDEX[key]
    return _gather_table(ia, flat), (None if ib is None else _gather_table(ib, flat))




def child_q4_tables(params):
    """Table of the quad-block pass A (pcgc_irn_child_q4, C = 16): [27][co 4][ci 16] = conv0_0.kernel[k][ci][co], then [co 4][ci 16] =
    conv1_0.kernel[0][ci][co] — a lane's operands of one (cell, child) pair are its output channel's 16 input channels, contiguous."""
    W00, W10 = params[0], params[4]
    C = W00.shape[1]
    return torch.cat([W00.detach().reshape(27, C, C // 4).permute(0, 2, 1).reshape(-1),
                      W10.detach().reshape(C, C // 4).t().reshape(-1)]).contiguous()




def irn_block_child(parent_nbr, x, params, tables, q4_table=None):
    """Fused InceptionResNet on a children level through the parent map (C = 16, 32); bit-identical to irn_block."""
    _f32(x, 'x')
    n_p = parent_nbr.shape[1]
    n, C = x.shape
    if n != 8 * n_p:
        raise PcgcError('irn_block_child: feature rows must be 8 x the parent level')
    ta, tb = tables
    t = torch.empty((n, C // 2), dtype=torch.float32, device=x.device)
    out = torch.empty((n, C), dtype=torch.float32, device=x.device)
    P = [p.data_ptr() for p in params]
    s = _stream(x)
    if PROFILE.counting:
        PROFILE.count_children(parent_nbr)
    tiles = (n_p + 15) // 16
    per_tile = {16: (416, 248), 32: (1216, 736)}[C]          # MFMA instructions per 16-parent tile (incl. the conv1_2 products of pass B)
    names = (f'k_child_irn_a<{C}>', f'k_child_irn_b<{C}>')
    q4 = PATH.CHILD_Q4 and C == 16 and q4_table is not None
    if q4:
        # per 16 parents, in units of 2048 flops: 224 groups x 16 4x4x1 instructions (512 flops each) per 64 parents + the 128 transposing ones per 128
        per_tile = ((216 + 8) * 16 // 4 // 4 + 128 // 8 // 4, per_tile[1])
        names = ('k_child_q4<0, 8, 2>', 'k_child_irn_b<16>')            # (pass B: the T2-gather instantiation of the packed-N kernel)
    forms = _irn_pass_formulas(n, C, 27 * n_p * 4, names)
    if q4:      # pass A in quad-block form writes t in its T2 layout; pass B is the packed-N kernel with T2 gather addresses (csrc/child_q4.hip)
        calls = (lambda: lib().pcgc_irn_child_q4(_p(parent_nbr), n_p, C, 1, _p(x), _ld(x), _p(q4_table), q4_table.numel() * 4, P[1], P[5], None,
                                                 None, 0, _p(t), C // 2, s),
                 lambda: lib().pcgc_irn_child_q4(_p(parent_nbr), n_p, C, 2, _p(t), C // 2, _p(tb), tb.numel() * 4, P[3], P[7], P[9], _p(x), _ld(x),
                                                 _p(out), C, s))
    else:
        calls = (lambda: lib().pcgc_irn_child_pass(_p(parent_nbr), n_p, C, 1, _p(x), _ld(x), _p(ta), ta.numel() * 4, P[1], P[5], None, None, 0,
                                                   _p(t), C // 2, s),
                 lambda: lib().pcgc_irn_child_pass(_p(parent_nbr), n_p, C, 2, _p(t), C // 2, _p(tb), tb.numel() * 4, P[3], P[7], P[9], _p(x), _ld(x),
                                                   _p(out), C, s))
    for (ps, name, bf, ff, comp), call, mf in zip(forms, calls, per_tile):
        prof = PROFILE.want((name, n))
        if prof:
            e0, e1 = PROFILE.bracket((name, n), name + ' (fused InceptionResNet pass on a children level, packed-N fp32 MFMA)', n, bf, ff,
                                     compulsory=comp, mfma_issued=tiles * mf * 2048)
            e0.record()
        check(call(), 'irn_child_pass')
        if prof:
            e1.record()
    return out


def conv_child(parent_nbr, x, table, bias, Cout, out=None, residual=None, relu=False):
    """k3 conv on the children level of `parent_nbr`'s level: x has 8 * n_parent rows."""
    _f32(x, 'x')
    n_p = parent_nbr.shape[1]
    if x.shape[0] != 8 * n_p:
        raise PcgcError('conv_child: feature rows must be 8 x the parent level')
    Cin = x.shape[1]
    if out is None:
        out = torch.empty((8 * n_p, Cout), dtype=torch.float32, device=x.device)
    res_p, res_ld = (None, 0) if residual is None else (_p(_f32(residual)), _ld(residual))
    n = 8 * n_p
    key = ('child_conv', Cin, Cout, n)
    prof = PROFILE.want(key)
    if prof:
        tiles = (n_p + 15) // 16
        per_tile = 64 * (Cin // 16) * 4 if Cout == 1 else 216 * (Cin // 16) * (Cout // 16) * 4        # MFMA instructions per 16-parent tile
        name = f'k_child_cls<{Cin // 16}>' if Cout == 1 else f'k_child_conv<{Cin // 16}, {Cout // 16}>'
        e0, e1 = PROFILE.bracket(key, name + f' (k3 {Cin}->{Cout} on a children level, parent-map halo gather + fp32 MFMA)', n,
                                 lambda P, a=Cin, b=Cout: P * a * 4 + P * 8 + n * b * 4, lambda P, a=Cin, b=Cout: 2 * P * a * b,
                                 compulsory=n * Cin * 4 + 27 * n_p * 4 + n * Cout * 4 + (0 if residual is None else n * Cout * 4),
                                 mfma_issued=tiles * per_tile * 2048)
        e0.record()
    check(lib().pcgc_conv_child(_p(parent_nbr), n_p, _p(x), Cin, _ld(x), _p(table), table.numel() * 4, _p(bias), res_p, res_ld,
                                int(relu), _p(out), Cout, _ld(out), _stream(x)), 'conv_child')
    if prof:
        e1.record()
    elif PROFILE.counting:
        PROFILE.count_children(parent_nbr)
    return out




def conv_rows(nbr, x, table, bias, Cout, out=None, residual=None, relu=False):
    """k3 conv 32 -> 32 on a plain level through its own map (csrc/rows_irn.hip: k_rows_conv); table = child_conv_table(kernel)."""
    _f32(x, 'x')
    n, Cin = x.shape
    if out is None:
        out = torch.empty((n, Cout), dtype=torch.float32, device=x.device)
    res_p, res_ld = (None, 0) if residual is None else (_p(_f32(residual)), _ld(residual))
    key = ('conv', Cin, Cout, n)
    prof = PROFILE.want(key)
    if prof:
        e0, e1 = PROFILE.bracket(key, f'k_rows_conv<{Cin // 16}, {Cout // 16}> (k3 {Cin}->{Cout} on a plain level, LDS-resident table + per-wave row ring, fp32 MFMA)', n,
                                 lambda P, a=Cin, b=Cout: P * a * 4 + P * 8 + n * b * 4, lambda P, a=Cin, b=Cout: 2 * P * a * b,
                                 compulsory=n * Cin * 4 + 27 * n * 4 + n * Cout * 4 + (0 if residual is None else n * Cout * 4),
                                 mfma_issued=2 * 27 * ((n + 15) // 16 * 16) * Cin * Cout)
        e0.record()
    check(lib().pcgc_conv_rows(_p(nbr), n, _p(x), Cin, _ld(x), _p(table), table.numel() * 4, _p(bias), res_p, res_ld, int(relu), _p(out), Cout,
                               _ld(out), _stream(x)), 'conv_rows')
    if prof:
        e1.record()
    elif PROFILE.counting:
        PROFILE.count(nbr)
    return out




def conv_packed64(nbr, x, table, bias, relu=False):
    """k3 conv 64 -> 64 on a level with its own map, present rows packed per workgroup tile and offset (pcgc_conv_packed64)."""
    _f32(x, 'x')
    n = x.shape[0]
    out = torch.empty((n, 64), dtype=torch.float32, device=x.device)
    key = ('conv', 64, 64, n)
    prof = PROFILE.want(key)
    if prof:
        # MFMA flops issued: the present pairs in packed 16-row tiles, plus about half a tile of padding per (workgroup tile of ~96 rows, offset)
        e0, e1 = PROFILE.bracket(key, 'k_conv_packed64 (k3 64->64, present-row packing: accumulators in LDS, B fragments in registers, fp32 MFMA)', n,
                                 lambda P, n=n: P * 64 * 4 + P * 8 + n * 64 * 4, lambda P: 2 * P * 64 * 64,
                                 compulsory=n * 64 * 4 + 27 * n * 4 + n * 64 * 4,
                                 mfma_issued=lambda P, n=n: 2 * (P + 8 * 27 * ((n + 95) // 96)) * 64 * 64)
        e0.record()
    check(lib().pcgc_conv_packed64(_p(nbr), n, _p(x), _ld(x), _p(table), table.numel() * 4, _p(bias), int(relu), _p(out), 64, _stream(x)),
          'conv_packed64')
    if prof:
        e1.record()
    elif PROFILE.counting:
        PROFILE.count(nbr)
    return out




def conv_down_rows(down, x, table, bias, Cout, relu=False):
    """k2 s2 down conv through the `down` map [8][n_coarse] (csrc/rows_irn.hip: k_rows_down); table = child_conv_table(kernel)."""
    _f32(x, 'x')
    n_in, Cin = x.shape
    n_c = down.shape[1]
    out = torch.empty((n_c, Cout), dtype=torch.float32, device=x.device)
    key = ('down', Cin, Cout, n_c)
    prof = PROFILE.want(key)
    if prof:      # k2 s2: every fine row is one (in,out) pair (bytes = N_fine Cin 4 + N_coarse Cout 4, flops = 2 N_fine Cin Cout)
        e0, e1 = PROFILE.bracket(key, f'k_rows_down<{Cin // 16}, {Cout // 16}> (k2 s2 {Cin}->{Cout} through the down map, LDS-resident table, fp32 MFMA)', n_c,
                                 lambda P, a=Cin, b=Cout: P * a * 4 + n_c * b * 4, lambda P, a=Cin, b=Cout: 2 * P * a * b,
                                 compulsory=n_in * Cin * 4 + 8 * n_c * 4 + n_c * Cout * 4,
                                 mfma_issued=2 * 8 * ((n_c + 15) // 16 * 16) * Cin * Cout, pairs=n_in)
        e0.record()
    check(lib().pcgc_conv_down_rows(_p(down), n_c, _p(x), n_in, Cin, _ld(x), _p(table), table.numel() * 4, _p(bias), int(relu), _p(out), Cout,
                                    _ld(out), _stream(x)), 'conv_down_rows')
    if prof:
        e1.record()
    return out


def conv_up2(x, W, bias, relu=False, rows=None):
    """MinkowskiGenerativeConvolutionTranspose k2 s2.  rows (int32 [n]): the input level is rows `rows` of x, read in place (a pruned
    level whose compacted features were never written); shapes without such a kernel gather the rows first."""
    _f32(x, 'x'); _f32(W, 'W')
    K, Cin, Cout = W.shape
    n_in = x.shape[0] if rows is None else rows.shape[0]
    key = ('up2', Cin, Cout, 8 * n_in)
    prof = PROFILE.want(key)
    if prof:      # generative transpose k2 s2: P = fine rows (bytes = N_fine (Cin + Cout) 4, flops = 2 N_fine Cin Cout)
        mfma = (Cin, Cout) in ((64, 32), (32, 16))
        e0, e1 = PROFILE.bracket(key, f'k_conv_up2<{Cin}, {Cout}> (generative transpose k2 s2, ' + ('eight GEMMs sharing the A fragments, fp32 MFMA)' if mfma else
                                 'one thread per (row, 16-byte chunk), VALU)'), 8 * n_in,
                                 lambda P, a=Cin, b=Cout: P * (a + b) * 4, lambda P, a=Cin, b=Cout: 2 * P * a * b,
                                 compulsory=n_in * Cin * 4 + 8 * n_in * Cout * 4 + (0 if rows is None else n_in * 4),
                                 mfma_issued=(2 * 8 * ((n_in + 15) // 16 * 16) * Cin * Cout) if mfma else None, pairs=8 * n_in)
        e0.record()
    try:
        if rows is not None:
            out = torch.empty((8 * rows.shape[0], Cout), dtype=torch.float32, device=x.device)
            rc = lib().pcgc_conv_up2_gather(rows.shape[0], _p(x), Cin, _ld(x), _p(rows), _p(W), _p(bias), int(relu), _p(out), Cout, _stream(x))
            if rc == 0:
                return out
            if rc != -3:
                check(rc, 'conv_up2_gather')
            x = gather_rows(x, rows)
        out = torch.empty((8 * x.shape[0], Cout), dtype=torch.float32, device=x.device)
        check(lib().pcgc_conv_up2(x.shape[0], _p(x), Cin, _ld(x), _p(W), _p(bias), int(relu), _p(out), Cout, _stream(x)), 'conv_up2')
        return out
    finally:
        if prof:
            e1.record()


# ------------------------------------------------------------------------------------------------ select / sort
def topk_mask(logits, k):
    """logits: [n,1] or [n] fp32 view -> uint8 mask [n] of the k largest (ties: lower row)."""
    _f32(logits, 'logits')
    n = logits.shape[0]
    ld = logits.stride(0)
    mask = torch.empty(n, dtype=torch.uint8, device=logits.device)
    ws_bytes = int(lib().pcgc_topk_workspace_bytes(n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=logits.device)
    check(lib().pcgc_topk_mask(_p(logits), ld, n, int(k), _p(mask), _p(ws), ws_bytes, _stream(logits)), 'topk_mask')
    return mask


def sort_zyx(coords, batch_major=False):
    """argsort by (z, y, x, batch) — array2vector's order — or, batch_major, by (batch, z, y, x): the items of a collated batch stay
    contiguous, each in the order it has when coded alone."""
    n = coords.shape[0]
    perm = torch.empty(n, dtype=torch.int32, device=coords.device)
    ws_bytes = int(lib().pcgc_sort_workspace_bytes(n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=coords.device)
    fn = lib().pcgc_sort_bzyx if batch_major else lib().pcgc_sort_zyx
    check(fn(_p(_i32(coords)), n, _p(perm), _p(ws), ws_bytes, _stream(coords)), 'sort_zyx')
    return perm


def batch_counts(coords):
    """rows per batch item (column 0), trailing empty items dropped -> list of ints (one device histogram + read-back)."""
    counts = torch.empty(16, dtype=torch.int32, device=coords.device)
    check(lib().pcgc_batch_counts(_p(_i32(coords)), coords.shape[0], _p(counts), _stream(coords)), 'batch_counts')
    c = counts.cpu().tolist()
    while len(c) > 1 and c[-1] == 0:
        c.pop()
    return c


def _i64_array(values):
    import ctypes
    return (ctypes.c_int64 * len(values))(*[int(v) for v in values])


def topk_mask_segments(logits, seg_rows, seg_k):
    """istopk per batch item (data_utils.py:77-89): item b = the next seg_rows[b] rows, keeps its seg_k[b] largest logits."""
    _f32(logits, 'logits')
    n = logits.shape[0]
    if sum(seg_rows) != n:
        raise PcgcError(f'topk_mask_segments: segments cover {sum(seg_rows)} of {n} rows')
    mask = torch.empty(n, dtype=torch.uint8, device=logits.device)
    ws_bytes = int(lib().pcgc_topk_workspace_bytes(n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=logits.device)
    check(lib().pcgc_topk_mask_segments(_p(logits), logits.stride(0), len(seg_rows), _i64_array(seg_rows), _i64_array(seg_k), _p(mask), _p(ws),
                                        ws_bytes, _stream(logits)), 'topk_mask_segments')
    return mask




def topk_select(logits, seg_rows, seg_k, coords=None, parent_coords=None, parent_stride=0):
    """prune_voxel in one sweep (autoencoder.py:239-249): per item b the seg_k[b] largest logits among its seg_rows[b] rows (istopk's tie
    rule) -> (bits uint8, wprefix int32: the rank bitmap of the candidate level; orig int32 [K]: the candidate row of every survivor;
    out_coords int32 [K, 4]).  The candidates' coordinates are `coords`, or derived from `parent_coords` (rows 8 i + j of a children
    level at parent_stride / 2)."""
    _f32(logits, 'logits')
    n = logits.shape[0]
    if sum(seg_rows) != n:
        raise PcgcError(f'topk_select: segments cover {sum(seg_rows)} of {n} rows')
    if len(seg_rows) > 16:
        raise PcgcError('topk_select: at most 16 segments')
    K = sum(int(min(max(k, 0), r)) for r, k in zip(seg_rows, seg_k))
    dev = logits.device
    words = (n + 63) // 64
    bits = torch.empty(words * 8, dtype=torch.uint8, device=dev)
    wprefix = torch.empty(words, dtype=torch.int32, device=dev)
    orig = torch.empty(K, dtype=torch.int32, device=dev)
    out = torch.empty((K, 4), dtype=torch.int32, device=dev)
    ws_bytes = int(lib().pcgc_topk_select_workspace_bytes(n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    check(lib().pcgc_topk_select(_p(logits), logits.stride(0), len(seg_rows), _i64_array(seg_rows), _i64_array(seg_k),
                                 _p(None if coords is None else _i32(coords)), _p(None if parent_coords is None else _i32(parent_coords)),
                                 int(parent_stride), _p(bits), _p(wprefix), _p(orig), _p(out), _p(ws), ws_bytes, _stream(logits)), 'topk_select')
    return bits, wprefix, orig, out


def kmap_k3_prune_sel(cand_nbr, bits, wprefix, orig):
    n_out = orig.shape[0]
    nbr = torch.empty((27, n_out), dtype=torch.int32, device=cand_nbr.device)
    check(lib().pcgc_kmap_k3_prune_sel(_p(cand_nbr), cand_nbr.shape[1], _p(bits), _p(wprefix), _p(orig), n_out, _p(nbr), _stream(cand_nbr)),
          'kmap_k3_prune_sel')
    return nbr


def kmap_k3_prune_parent_sel(parent_nbr, bits, wprefix, orig):
    """k3 map of a pruned children level from the PARENT level's map and the rank bitmap of ops.topk_select."""
    n_out = orig.shape[0]
    nbr = torch.empty((27, n_out), dtype=torch.int32, device=parent_nbr.device)
    check(lib().pcgc_kmap_k3_prune_parent_sel(_p(parent_nbr), parent_nbr.shape[1], _p(bits), _p(wprefix), _p(orig), n_out, _p(nbr),
                                              _stream(parent_nbr)), 'kmap_k3_prune_parent_sel')
    return nbr


def gather_rows(feats, orig):
    """out[r] = feats[orig[r]] (feats: 2-D row-major view with a leading dimension; channels % 4 == 0)."""
    _f32(feats)
    C = feats.shape[1]
    out = torch.empty((orig.shape[0], C), dtype=torch.float32, device=feats.device)
    check(lib().pcgc_gather_rows_f32_ld(_p(feats), C, _ld(feats), _p(orig), orig.shape[0], _p(out), _stream(feats)), 'gather_rows_f32_ld')
    return out


def gather_coords(coords, perm):
    out = torch.empty_like(coords)
    check(lib().pcgc_gather_rows_i32x4(_p(_i32(coords)), _p(perm), coords.shape[0], _p(out), _stream(coords)), 'gather_rows_i32x4')
    return out


def gather_feats(feats, perm):
    feats = _f32(feats).contiguous()
    out = torch.empty_like(feats)
    check(lib().pcgc_gather_rows_f32(_p(feats), feats.shape[1], _p(perm), feats.shape[0], _p(out), _stream(feats)), 'gather_rows_f32')
    return out


# ------------------------------------------------------------------------------------------------ pointwise (ME-style graphs)
def relu(x, inplace=False):
    x = _f32(x).contiguous()
    out = x if inplace else torch.empty_like(x)
    check(lib().pcgc_relu(_p(x), x.numel(), _p(out), _stream(x)), 'relu')
    return out


def add(a, b):
    a, b = _f32(a).contiguous(), _f32(b).contiguous()
    if a.shape != b.shape:
        raise PcgcError(f'add: shapes {tuple(a.shape)} and {tuple(b.shape)} differ')
    out = torch.empty_like(a)
    check(lib().pcgc_add(_p(a), _p(b), a.numel(), _p(out), _stream(a)), 'add')
    return out


# ------------------------------------------------------------------------------------------------ entropy
def round_minmax(feats):
    feats = _f32(feats).contiguous()
    mm = torch.empty(2, dtype=torch.float32, device=feats.device)
    check(lib().pcgc_round_minmax(_p(feats), feats.numel(), _p(mm), _stream(feats)), 'round_minmax')
    return mm


def symbolize(feats, min_v):
    feats = _f32(feats).contiguous()
    sym = torch.empty(feats.shape, dtype=torch.int16, device=feats.device)
    check(lib().pcgc_symbolize(_p(feats), feats.numel(), float(min_v), _p(sym), _stream(feats)), 'symbolize')
    return sym


def desymbolize(sym, min_v):
    out = torch.empty(sym.shape, dtype=torch.float32, device=sym.device)
    check(lib().pcgc_desymbolize(_p(_dev(sym, torch.int16, 'sym')), sym.numel(), float(min_v), _p(out), _stream(sym)), 'desymbolize')
    return out


def quantize_symbols(feats):
    """-> (min_v, max_v as np.float32, sym int16 ndarray of feats.shape): round + symbol range + symbolise on the device,
    ONE synchronising device->host copy of [min | max | symbols]."""
    feats = _f32(feats).contiguous()
    n = feats.numel()
    buf = torch.empty(4 + n, dtype=torch.int16, device=feats.device)          # [minmax as 2 fp32 = 4 int16 | sym]
    check(lib().pcgc_quantize_symbols(_p(feats), n, _p(buf), buf.data_ptr() + 8, _stream(feats)), 'quantize_symbols')
    host = buf.cpu().numpy()
    mm = host[:4].view(np.float32)
    return np.float32(mm[0]), np.float32(mm[1]), host[4:].reshape(feats.shape)


def quantize_symbols_segments(feats, seg_rows):
    """quantize_symbols per batch item (contiguous row segments), ONE synchronising copy
    -> ([(min_v, max_v)] per item as np.float32, sym int16 ndarray of feats.shape)."""
    feats = _f32(feats).contiguous()
    n, C = feats.shape
    B = len(seg_rows)
    if sum(seg_rows) != n:
        raise PcgcError(f'quantize_symbols_segments: segments cover {sum(seg_rows)} of {n} rows')
    buf = torch.empty(4 * B + n * C, dtype=torch.int16, device=feats.device)          # [B x (min, max) fp32 | symbols]
    check(lib().pcgc_quantize_symbols_segments(_p(feats), C, B, _i64_array(seg_rows), _p(buf), buf.data_ptr() + 8 * B, _stream(feats)),
          'quantize_symbols_segments')
    host = buf.cpu().numpy()
    mm = host[:4 * B].view(np.float32).reshape(B, 2)
    return [(np.float32(a), np.float32(b)) for a, b in mm], host[4 * B:].reshape(n, C)


def cdf_table(params, C, min_v, max_v):
    """-> (cdf_u16 as int16-typed tensor [C, L+1] holding the uint16 bit patterns, cdf_f32 [C, L+1])."""
    L = int(max_v - min_v) + 1
    q = torch.empty((C, L + 1), dtype=torch.int16, device=params.device)
    f = torch.empty((C, L + 1), dtype=torch.float32, device=params.device)
    check(lib().pcgc_cdf_table(_p(_f32(params, 'params')), C, float(min_v), float(max_v), _p(q), _p(f), _stream(params)), 'cdf_table')
    return q, f


# ------------------------------------------------------------------------------------------------ D1 metric
_D1_OFFSETS = {}


def _d1_offsets(device, radius=12):
    """(dx,dy,dz,d2) for every lattice offset with max|d| <= radius, sorted by d2 (ties in a fixed order)."""
    key = (str(device), radius)
    if key not in _D1_OFFSETS:
        r = np.arange(-radius, radius + 1)
        g = np.stack(np.meshgrid(r, r, r, indexing='ij'), -1).reshape(-1, 3)
        d2 = (g * g).sum(1)
        order = np.lexsort((g[:, 0], g[:, 1], g[:, 2], d2))
        tab = np.concatenate([g[order], d2[order, None]], 1).astype(np.int32)
        # an offset at Chebyshev distance > radius could be closer than the corner offsets: keep only d2 <= radius^2
        tab = tab[tab[:, 3] <= radius * radius]
        _D1_OFFSETS[key] = torch.from_numpy(np.ascontiguousarray(tab)).to(device)
    return _D1_OFFSETS[key]


_D1_CELL_OFFSETS = {}


def _d1_cell_offsets(device, reach_cells=4):
    """(ox, oy, oz, lower bound of the squared distance) for every CELL offset with max|o| <= reach_cells, ascending in the bound: a voxel of the
    query's own cell and a voxel of the cell at offset o are at least max(0, 4 |o| - 3) apart per axis."""
    key = (str(device), reach_cells)
    if key not in _D1_CELL_OFFSETS:
        r = np.arange(-reach_cells, reach_cells + 1)
        g = np.stack(np.meshgrid(r, r, r, indexing='ij'), -1).reshape(-1, 3)
        gap = np.maximum(0, 4 * np.abs(g) - 3)
        lb = (gap * gap).sum(1)
        order = np.lexsort((g[:, 0], g[:, 1], g[:, 2], lb))
        tab = np.concatenate([g[order], lb[order, None]], 1).astype(np.int32)
        _D1_CELL_OFFSETS[key] = torch.from_numpy(np.ascontiguousarray(tab)).to(device)
    return _D1_CELL_OFFSETS[key]


def d1_nn(a, b, radius=12):
    """a, b: int32 [N,4] device coordinate tensors (stride 1).  -> (sum of squared NN distances a->b, max, unresolved count)."""
    s = torch.empty(1, dtype=torch.float64, device=a.device)
    m = torch.empty(1, dtype=torch.int64, device=a.device)
    u = torch.empty(1, dtype=torch.int32, device=a.device)
    if PATH.D1_CELLS and b.shape[0] > 0:
        # cloud B as stride-4 cells: hash of the cells + a 64-bit occupancy mask per cell; reach: cells up to 4 away, i.e. every voxel nearer
        # than 4 * 5 - 3 = 17 has been seen when the search ends
        reach_cells = max(1, (int(radius) + 3) // 4 + 1)
        # (the hash of the quantised rows maps a cell to the FIRST row that lies in it: that row's slot of `masks` is the cell's mask)
        table = HashTable(coords_quantize(_i32(b), 4), 4)
        masks = torch.empty(b.shape[0], dtype=torch.int64, device=b.device)
        check(lib().pcgc_d1_cell_masks(_p(b), b.shape[0], _p(table.keys), _p(table.vals), table.cap, _p(masks), b.shape[0], _stream(b)),
              'd1_cell_masks')
        off = _d1_cell_offsets(a.device, reach_cells)
        reach = 4 * (reach_cells + 1) - 3
        check(lib().pcgc_d1_nn_cells(_p(_i32(a)), a.shape[0], _p(table.keys), _p(table.vals), table.cap, _p(masks), _p(off), off.shape[0],
                                     reach * reach, _p(s), _p(m), _p(u), _stream(a)), 'd1_nn_cells')
        return s, m, u
    table = HashTable(_i32(b), 1)
    off = _d1_offsets(a.device, radius)
    check(lib().pcgc_d1_nn(_p(_i32(a)), a.shape[0], _p(table.keys), _p(table.vals), table.cap, _p(off), off.shape[0], _p(s), _p(m),
                           _p(u), _stream(a)), 'd1_nn')
    return s, m, u


# ------------------------------------------------------------------------------------------------ host codecs (numpy)
def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def set_rc_impl(impl):
    """0 automatic, 1 portable scalar range decoder (bit-identical; for A/B tests)."""
    check(lib().pcgc_set_rc_impl(int(impl)), 'set_rc_impl')


RC_CKPT_WORDS = 6            # include/pcgc_hip.h PCGC_RC_CKPT_WORDS


def set_rc_threads(threads):
    """Threads of the indexed range decoder (0 = min(8, hardware threads))."""
    check(lib().pcgc_set_rc_threads(int(threads)), 'set_rc_threads')


def set_rc_lanes(mode):
    """Lane-parallel indexed range decoder (eight segments per 512-bit register on the calling thread): -1 automatic (thread budget of one
    or two), 0 never, 1 always.  Same symbols either way."""
    check(lib().pcgc_set_rc_lanes(int(mode)), 'set_rc_lanes')


def crc32(data, crc=0):
    """zlib.crc32(data, crc), by the library's carry-less-multiply fold (the value the native item files carry in their sidecars)"""
    src = np.frombuffer(data, np.uint8)
    return int(lib().pcgc_crc32(int(crc) & 0xFFFFFFFF, src.ctypes.data if src.size else None, src.size))


def rc_encode(cdf_u16, sym, checkpoints=0):
    """-> stream bytes, or (stream bytes, index uint32 [checkpoints, RC_CKPT_WORDS]) when checkpoints > 0: the decoder state at
    evenly spread row boundaries, for rc_decode(index=...).  The stream is the same either way."""
    cdf = _np(cdf_u16, np.uint16)
    sym = _np(sym, np.int16).ravel()
    C, Lp = cdf.shape
    cap = sym.size * 2 + 64
    index = np.zeros((int(checkpoints), RC_CKPT_WORDS), np.uint32) if checkpoints > 0 else None
    while True:
        buf = np.empty(cap, np.uint8)
        if index is None:
            n = int(lib().pcgc_rc_encode(cdf.ctypes.data, C, Lp, sym.ctypes.data, sym.size, buf.ctypes.data, cap))
        else:
            n = int(lib().pcgc_rc_encode_indexed(cdf.ctypes.data, C, Lp, sym.ctypes.data, sym.size, buf.ctypes.data, cap,
                                                 index.shape[0], index.ctypes.data))
        if n >= 0:
            return buf[:n].tobytes() if index is None else (buf[:n].tobytes(), index)
        if n == -(2 ** 63):
            raise PcgcError('rc_encode: symbol outside the CDF table')
        cap = -n


def rc_decode(cdf_u16, data, n, index=None):
    """`index` (optional): the checkpoints rc_encode(..., checkpoints=k) returned for this stream -> segments decoded in parallel."""
    cdf = _np(cdf_u16, np.uint16)
    C, Lp = cdf.shape
    src = np.frombuffer(data, np.uint8)
    out = np.empty(n, np.int16)
    if index is None:
        check(lib().pcgc_rc_decode(cdf.ctypes.data, C, Lp, src.ctypes.data, src.size, out.ctypes.data, n), 'rc_decode')
    else:
        idx = np.ascontiguousarray(index, dtype=np.uint32).reshape(-1, RC_CKPT_WORDS)
        check(lib().pcgc_rc_decode_indexed(cdf.ctypes.data, C, Lp, src.ctypes.data, src.size, out.ctypes.data, n, idx.shape[0],
                                           idx.ctypes.data), 'rc_decode_indexed')
    return out


def _stems(stems):
    """file stems as a C array of byte strings (+ the Python objects that own the bytes)"""
    import ctypes
    enc = [_os.fsencode(s) for s in stems]
    return (ctypes.c_char_p * len(enc))(*enc), enc


def _table_fn():
    """address of pcgc_reference_table (libpcgc_reftable.so): the CDF table in the reference's arithmetic, callable from native threads"""
    import ctypes
    from ._lib import reftable_lib
    return ctypes.cast(reftable_lib().pcgc_reference_table, ctypes.c_void_p)


_WARM_SCRATCH = np.zeros((64, 16), np.uint16)


def table_warm(eb_params, C):
    """Run the reference-arithmetic CDF-table evaluation (libpcgc_reftable.so) once on a small dummy range and throw the result away.
    The real evaluation sits on the critical path of every encode, right after the sync that hands over the symbol range, and its ~60
    ATen operators run 2-2.5x slower from cold instruction / data caches (0.36 ms against 0.15 ms hot on the bench host) than from warm
    ones.  Called while the host would otherwise wait for the GPU, it changes no result and memoises nothing — the table the encoder
    codes with is still evaluated for its own range, after the range is known."""
    from ._lib import reftable_lib
    P = _np(eb_params, np.float32)
    if P.size != 44 * C or C > _WARM_SCRATCH.shape[0]:
        return
    reftable_lib().pcgc_reference_table(P.ctypes.data, int(C), -7.0, 7.0, _WARM_SCRATCH.ctypes.data, None)


def items_encode(stems, sym_h, xyz8, rows, ranges, counts, eb_params, index_segments, write_coords=True, threads=0):
    """Write the bitstream files of every item (native threads): sym_h int16 [sum rows, C], xyz8 int32 [sum rows, 3] (stride-8
    coordinates / 8), rows / ranges [(min_v, max_v)] / counts [(N4, N2, N1)] per item, eb_params = the 44*C packed entropy parameters."""
    sym = _np(sym_h, np.int16)
    C = sym.shape[1]
    xyz = _np(xyz8, np.int32)
    arr, keep = _stems(stems)
    r = np.asarray(rows, np.int64)
    rg = _np(ranges, np.float32).reshape(-1, 2)
    ct = _np(counts, np.int32).reshape(-1, 3)
    P = _np(eb_params, np.float32)
    if P.size != 44 * C:
        raise PcgcError(f'items_encode: {P.size} entropy parameters for {C} channels (44 per channel)')
    check(lib().pcgc_items_encode(len(stems), arr, sym.ctypes.data, xyz.ctypes.data, r.ctypes.data, rg.ctypes.data, C, ct.ctypes.data, P.ctypes.data,
                                  _table_fn(), int(index_segments), int(write_coords), int(threads)), 'items_encode')


def items_probe(stems):
    """-> (rows int64 [n], C, ranges float32 [n, 2], counts int32 [n, 3], native_coords int32 [n]) from the files of every item."""
    import ctypes
    n = len(stems)
    arr, keep = _stems(stems)
    rows, ranges, counts, native = np.zeros(n, np.int64), np.zeros((n, 2), np.float32), np.zeros((n, 3), np.int32), np.zeros(n, np.int32)
    C = ctypes.c_int32(0)
    check(lib().pcgc_items_probe(n, arr, rows.ctypes.data, ctypes.addressof(C), ranges.ctypes.data, counts.ctypes.data, native.ctypes.data), 'items_probe')
    return rows, int(C.value), ranges, counts, native


def items_decode(stems, rows, C, ranges, native, eb_params, use_sidecar=True, threads=0, level_scale=0, level_out=None):
    """-> (sym int16 [sum rows, C], coordinates) decoded from the files of every item (native threads).  Coordinates: xyz8 int32
    [sum rows, 3] in stream order, or with level_scale = s > 0 the sorted coordinate level int32 [sum rows, 4] = (item, s x, s y, s z),
    every item in (z, y, x) order — written into `level_out` (e.g. a pinned buffer's numpy view, at least [sum rows, 4]) if given."""
    arr, keep = _stems(stems)
    total = int(np.sum(rows))
    sym = np.empty((total, C), np.int16)
    if level_scale:
        if level_out is None:
            level_out = np.empty((total, 4), np.int32)
        if level_out.dtype != np.int32 or level_out.ndim != 2 or level_out.shape[1] != 4 or level_out.shape[0] < total or not level_out.flags.c_contiguous:
            raise PcgcError('items_decode: level_out must be a C-contiguous int32 [>= rows, 4] array')
        xyz = level_out
    else:
        xyz = np.empty((total, 3), np.int32)
    P = _np(eb_params, np.float32)
    if P.size != 44 * C:
        # (the library reads 44 * C floats: a channel count taken from a damaged `_H.bin` must never reach it)
        raise PcgcError(f'items_decode: {P.size} entropy parameters for {C} channels (44 per channel)')
    rc = lib().pcgc_items_decode(len(stems), arr, rows.ctypes.data, C, ranges.ctypes.data, native.ctypes.data, P.ctypes.data, _table_fn(), int(use_sidecar),
                                 sym.ctypes.data, xyz.ctypes.data, 1 if level_scale else 0, int(level_scale) if level_scale else 1, int(threads))
    check(rc, 'items_decode')
    return sym, xyz[:total]


def frame_decode(stem, C, eb_params, sym_buf, level_buf, use_sidecar=True, level_scale=8, threads=2):
    """One cloud's files -> symbols and sorted coordinate level, in ONE library call, into the caller's (pinned) numpy buffers sym_buf
    int16 [cap, C] and level_buf int32 [cap, 4].  -> (rows, (min_v, max_v), (N4, N2, N1), native_coords), or (rows_needed, None, None,
    None) when the buffers are too small (nothing decoded: grow them and call again)."""
    import ctypes
    cap = min(sym_buf.shape[0], level_buf.shape[0])
    if sym_buf.dtype != np.int16 or level_buf.dtype != np.int32 or sym_buf.shape[1:] != (C,) or level_buf.shape[1:] != (4,) \
            or not sym_buf.flags.c_contiguous or not level_buf.flags.c_contiguous:
        raise PcgcError('frame_decode: buffers must be C-contiguous int16 [cap, C] and int32 [cap, 4]')
    info = (ctypes.c_int64 * 6)()
    rng = (ctypes.c_float * 2)()
    P = _np(eb_params, np.float32)
    if P.size != 44 * int(C):
        raise PcgcError(f'frame_decode: {P.size} entropy parameters for {C} channels (44 per channel)')
    rc = lib().pcgc_frame_decode(_os.fsencode(stem), int(C), P.ctypes.data, _table_fn(), int(bool(use_sidecar)), int(level_scale), cap,
                                 sym_buf.ctypes.data, level_buf.ctypes.data, info, rng, int(threads))
    if rc == 1:
        return int(info[0]), None, None, None
    check(rc, 'frame_decode')
    return int(info[0]), (np.float32(rng[0]), np.float32(rng[1])), (int(info[2]), int(info[3]), int(info[4])), bool(info[5])


def frame_decode_begin(stem, C, eb_params, sym_buf, level_buf, use_sidecar=True, level_scale=8, threads=2):
    """frame_decode in two halves: returns as soon as the coordinate level is in level_buf (the feature stream keeps decoding on the
    library's threads) -> the same tuple as frame_decode; the symbols in sym_buf are valid only after frame_decode_end().  When the
    buffers are too small (rows_needed, None, None, None) nothing is pending."""
    import ctypes
    cap = min(sym_buf.shape[0], level_buf.shape[0])
    if sym_buf.dtype != np.int16 or level_buf.dtype != np.int32 or sym_buf.shape[1:] != (C,) or level_buf.shape[1:] != (4,) \
            or not sym_buf.flags.c_contiguous or not level_buf.flags.c_contiguous:
        raise PcgcError('frame_decode: buffers must be C-contiguous int16 [cap, C] and int32 [cap, 4]')
    info = (ctypes.c_int64 * 6)()
    rng = (ctypes.c_float * 2)()
    P = _np(eb_params, np.float32)
    if P.size != 44 * int(C):
        raise PcgcError(f'frame_decode: {P.size} entropy parameters for {C} channels (44 per channel)')
    rc = lib().pcgc_frame_decode_begin(_os.fsencode(stem), int(C), P.ctypes.data, _table_fn(), int(bool(use_sidecar)), int(level_scale), cap,
                                       sym_buf.ctypes.data, level_buf.ctypes.data, info, rng, int(threads))
    if rc == 1:
        return int(info[0]), None, None, None
    check(rc, 'frame_decode')
    return int(info[0]), (np.float32(rng[0]), np.float32(rng[1])), (int(info[2]), int(info[3]), int(info[4])), bool(info[5])


def frame_decode_end():
    """wait for the feature stream of the frame frame_decode_begin started on this thread (raises what frame_decode would have raised)"""
    check(lib().pcgc_frame_decode_end(), 'frame_decode')


def set_oct_tiled(on):
    """Coordinate codec: groups of subtrees coded independently (1, the default for clouds of >= 8192 points; n > 1: that many
    groups) or always one stream (0).  A/B tests."""
    check(lib().pcgc_set_oct_tiled(int(on)), 'set_oct_tiled')


def set_oct_model(model):
    """context model of the octree ENCODER: 1 = stream versions 4 / 5 (mixed-shape prior, fast start; default), 0 = versions 2 / 3"""
    check(lib().pcgc_set_oct_model(int(model)), 'set_oct_model')


def oct_encode(xyz):
    xyz = _np(xyz, np.int32)
    if xyz.ndim != 2 or xyz.shape[1] != 3:
        raise PcgcError('oct_encode: expected [n,3] coordinates')
    cap = 64 + 4 * len(xyz)
    while True:
        buf = np.empty(cap, np.uint8)
        n = int(lib().pcgc_oct_encode(xyz.ctypes.data, len(xyz), buf.ctypes.data, cap))
        if n >= 0:
            return buf[:n].tobytes()
        if n == -(2 ** 63):
            raise PcgcError('oct_encode: coordinates must be in [0, 2^21)')
        cap = -n


def oct_decode(data):
    src = np.frombuffer(data, np.uint8)
    n = int(lib().pcgc_oct_decode_count(src.ctypes.data, src.size))
    if n < 0:
        raise PcgcError('oct_decode: not a PCGO stream')
    out = np.empty((n, 3), np.int32)
    check(lib().pcgc_oct_decode(src.ctypes.data, src.size, out.ctypes.data, n), 'oct_decode')
    return out

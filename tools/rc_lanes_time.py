#!/usr/bin/env python3
"""Indexed range decoder: segment-pool form (one or two segments per pool thread) against the lane-parallel form (eight segments per zmm
register on the calling thread, pcgc_set_rc_lanes) on a vox10-sized latent (149 856 symbols, 21-symbol alphabet, 16 checkpoints)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pcgcv2_amd import ops
from oracle import pcgc_oracle as orc
rng = np.random.default_rng(0)
L, n = 21, 18732
pmf = np.exp(-0.5 * ((np.arange(L) - 10) / 2.2) ** 2)[None, :].repeat(8, 0) + 1e-4
cdf = np.concatenate([np.zeros((8, 1)), np.cumsum(pmf / pmf.sum(1, keepdims=True), 1)], 1).clip(0, 1)
table = orc.cdf_u16(cdf.astype(np.float32))
p = pmf / pmf.sum(1, keepdims=True)
sym = np.stack([rng.choice(L, size=n, p=p[c]) for c in range(8)], 1).astype(np.int16)
data, index = ops.rc_encode(table, sym, checkpoints=16)
print('symbols', sym.size, 'bytes', len(data), 'cpus', len(os.sched_getaffinity(0)))


def t(f, reps=300):
    for _ in range(30): f()
    ts = []
    for _ in range(reps):
        a = time.perf_counter(); f(); ts.append(time.perf_counter() - a)
    return np.median(ts) * 1e3


for th in (1, 2, 4, 8):
    ops.set_rc_threads(th)
    for lanes in (0, 1):
        ops.set_rc_lanes(lanes)
        out = ops.rc_decode(table, data, sym.size, index=index)
        assert np.array_equal(out, sym.ravel())
        print(f'threads {th} lanes {lanes}: {t(lambda: ops.rc_decode(table, data, sym.size, index=index)):.3f} ms')
ops.set_rc_lanes(-1); ops.set_rc_threads(0)
print('serial (no index): %.3f ms' % t(lambda: ops.rc_decode(table, data, sym.size), 50))

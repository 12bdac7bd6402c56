#!/usr/bin/env python3
"""Time of the native coordinate codec (octree streams, groups coded side by side) on the stride-8 level of shell10, by thread count."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pcgcv2_amd import ops, synthetic
xyz = np.unique(synthetic.shell('shell10').numpy() // 8, axis=0).astype(np.int32)
def med(fn, n=200):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return statistics.median(ts) * 1e3
print(len(xyz), 'voxels')
for groups in (1, 8, 16):
    ops.set_oct_tiled(groups)
    for thr in (1, 2, 4, 8):
        ops.set_rc_threads(thr)
        s = ops.oct_encode(xyz)
        print(f'groups {groups:2d} threads {thr}: {len(s)} bytes  encode {med(lambda: ops.oct_encode(xyz)):.3f} ms  decode {med(lambda: ops.oct_decode(s)):.3f} ms')
ops.set_rc_threads(0); ops.set_oct_tiled(1)

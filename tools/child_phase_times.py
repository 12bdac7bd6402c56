#!/usr/bin/env python3
"""Per-phase shader-clock cycles of the children-level kernels on the stride-1 (C = 16) candidates of shell10, per tile and wave:
prologue (map loads + their latency), cell loop (gathers + MFMAs), epilogue issue (LDS staging + stores issued), drain (the wait for the
tile's stores: in the product that wait is the next tile's prologue).  Needs the timing build:
    PCGC_BUILD_VARIANT=timing PCGC_EXTRA_HIPCC_FLAGS=-DPCGC_CHILD_TIMING python -m pcgcv2_amd._build
    PCGC_LIB=pcgcv2_amd/libpcgc_hip_timing.so python tools/child_phase_times.py [tuning codes ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd._lib import lib, LIB_PATH
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
dev = torch.device('cuda:0')
L = ctypes.CDLL(LIB_PATH)
C = 16
codes = [int(a) for a in sys.argv[1:]] or [0]


def read(fn):
    buf = (ctypes.c_ulonglong * 8)()
    getattr(L, fn)(buf, 1)
    return list(buf)


pts = synthetic.shell('shell10', device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
parent = CoordMap(c4, 1, unique=True).build_pyramid(1)
pk = parent.k3; n_p = len(parent); n = 8 * n_p
x = torch.randn((n, C), device=dev)
blk = InceptionResNet(C).to(dev)
params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
tabs = ops.child_irn_tables(params)
P = [p.data_ptr() for p in params]
W = torch.randn((27, C, C), device=dev) * 0.05; b = torch.randn((1, C), device=dev)
tab = ops.child_conv_table(W); tc = ops.child_cls_table(torch.randn((27, C, 1), device=dev) * 0.05)
s = torch.cuda.current_stream().cuda_stream
t = torch.empty((n, C // 2), device=dev); out = torch.empty((n, C), device=dev)
runs = {
    'pass A': (lambda code: 'pcgc_child_timing_a16_mt2' if code >= 200 else 'pcgc_child_timing_a16',
               lambda: lib().pcgc_irn_child_pass(pk.data_ptr(), n_p, C, 1, x.data_ptr(), C, tabs[0].data_ptr(), tabs[0].numel() * 4, P[1], P[5], None, None, 0, t.data_ptr(), C // 2, s)),
    'pass B': (lambda code: 'pcgc_child_timing_b16_mt2' if code >= 200 else 'pcgc_child_timing_b16',
               lambda: lib().pcgc_irn_child_pass(pk.data_ptr(), n_p, C, 2, t.data_ptr(), C // 2, tabs[1].data_ptr(), tabs[1].numel() * 4, P[3], P[7], P[9], x.data_ptr(), C, out.data_ptr(), C, s)),
    'conv': (lambda code: 'pcgc_child_timing', lambda: ops.conv_child(pk, x, tab, b, C)),
    'cls': (lambda code: 'pcgc_child_timing', lambda: ops.conv_child(pk, x, tc, b[:, :1].contiguous(), 1)),
}
print(f'{n} rows, {n_p} parents; cycles per tile and wave (a tile = 16 parents, or 32 with two M tiles per wave)')
for code in codes:
    ops.set_child_tuning(code, 0)
    for name, (reader, run) in runs.items():
        fn = reader(code)
        for _ in range(2): run()
        read(fn)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        v = read(fn)
        tiles = max(v[3], 1)
        pro, loop, drain, tot = v[0] / tiles, v[1] / tiles, v[5] / tiles, v[4] / tiles
        print(f'code {code:4d} {name:7s} {e0.elapsed_time(e1) / 5 * 1e3:7.1f} us | prologue {pro:7.0f}  cell loop {loop:7.0f}  epilogue issue {tot - pro - loop - drain:7.0f}  '
              f'store drain {drain:7.0f}  total {tot:7.0f}  ({tiles // 5} tiles per launch)', flush=True)
ops.set_child_tuning(0, 0)

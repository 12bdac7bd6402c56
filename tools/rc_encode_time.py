#!/usr/bin/env python3
"""Time of the range encoder (with and without the decoding index) and decoder on the real latent symbols of shell10."""
import os, sys, tempfile, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcgcv2_amd
pcgcv2_amd.configure_host_threads()
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
coords = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
model = PCCModel().to(dev); model.load_state_dict(synthetic.synthetic_state_dict())
coder = Coder(model, os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'f'))
x = SparseTensor(torch.ones((len(pts), 1), device=dev), coordinates=coords, tensor_stride=1, device=dev)
y = coder.encode(x)
eb = model.entropy_bottleneck
min_v, max_v, sym_h = ops.quantize_symbols(y.F.contiguous())
tab = eb.host_table(min_v, max_v, dev)
def med(fn, n=60):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return statistics.median(ts) * 1e3
s, idx = ops.rc_encode(tab, sym_h, checkpoints=8)
s16, idx16 = ops.rc_encode(tab, sym_h, checkpoints=16)
print(f'symbols {sym_h.size}  bytes {len(s)}  alphabet {tab.shape[1] - 1}')
print(f'rc_encode          {med(lambda: ops.rc_encode(tab, sym_h)):.3f} ms')
print(f'rc_encode indexed  {med(lambda: ops.rc_encode(tab, sym_h, checkpoints=8)):.3f} ms')
print(f'rc_decode serial   {med(lambda: ops.rc_decode(tab, s, sym_h.size)):.3f} ms')
print(f'rc_decode indexed 8   {med(lambda: ops.rc_decode(tab, s, sym_h.size, index=idx)):.3f} ms')
print(f'rc_decode indexed 16  {med(lambda: ops.rc_decode(tab, s16, sym_h.size, index=idx16)):.3f} ms')
for thr in (1, 2, 4):
    ops.set_rc_threads(thr)
    print(f'  {thr} threads: 8 -> {med(lambda: ops.rc_decode(tab, s, sym_h.size, index=idx)):.3f} ms, 16 -> {med(lambda: ops.rc_decode(tab, s16, sym_h.size, index=idx16)):.3f} ms')
ops.set_rc_threads(0)

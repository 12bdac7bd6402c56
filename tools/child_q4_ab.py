#!/usr/bin/env python3
"""Quad-block pass A (round 5, csrc/child_q4.h) against the packed-N pass A: time and bit-for-bit comparison on the stride-1 (C = 16)
candidates of a cloud (default shell10: 2 045 536 rows), plus pass B / conv / cls as controls."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd._lib import lib
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
dev = torch.device('cuda:0')
C = 16
name = sys.argv[1] if len(sys.argv) > 1 else 'shell10'
pts = synthetic.shell(name, device=dev) if name in synthetic.SHELLS else synthetic.cloud(name, device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
parent = CoordMap(c4, 1, unique=True).build_pyramid(1)
pk = parent.k3
n_p = len(parent); n = 8 * n_p
g = torch.Generator(device='cpu').manual_seed(0)
x = torch.randn((n, C), generator=g).to(dev)
blk = InceptionResNet(C).to(dev)
params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
with torch.no_grad():
    for p_ in params: p_.normal_(0, 0.1)
tabs = ops.child_irn_tables(params)
tq = ops.child_q4_tables(params)
P = [p.data_ptr() for p in params]
s = torch.cuda.current_stream().cuda_stream
t = torch.empty((n, C // 2), device=dev); t2 = torch.full((n, C // 2), -7.0, device=dev); out = torch.empty((n, C), device=dev)
def pass_a(): ops.check(lib().pcgc_irn_child_pass(pk.data_ptr(), n_p, C, 1, x.data_ptr(), C, tabs[0].data_ptr(), tabs[0].numel() * 4, P[1], P[5], None, None, 0, t.data_ptr(), C // 2, s), 'a')
def pass_q(): ops.check(lib().pcgc_irn_child_q4(pk.data_ptr(), n_p, C, 1, x.data_ptr(), C, tq.data_ptr(), tq.numel() * 4, P[1], P[5], None, None, 0, t2.data_ptr(), C // 2, s), 'q')
def pass_b(): ops.check(lib().pcgc_irn_child_pass(pk.data_ptr(), n_p, C, 2, t.data_ptr(), C // 2, tabs[1].data_ptr(), tabs[1].numel() * 4, P[3], P[7], P[9], x.data_ptr(), C, out.data_ptr(), C, s), 'b')
Wc = (torch.randn((27, C, 1), generator=g) * 0.05).to(dev); bc = torch.randn((1, 1), generator=g).to(dev)
tcp, tcq = ops.child_cls_table(Wc), ops.child_q4_cls_table(Wc)
def cls_p(): return ops.conv_child(pk, x, tcp, bc, 1)
def cls_q(): return ops.cls_child_q4(pk, x, tcq, bc)
def med(f, reps=15):
    for _ in range(3): f()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts), min(ts)
print(f'{name}: {n} rows ({n_p} parents), C = {C}')
pass_a(); pass_q(); torch.cuda.synchronize()
eq = torch.equal(t, t2)
print('quad-block == packed-N:', eq)
if not eq:
    d = (t != t2)
    print('  differing elements:', int(d.sum()), 'of', d.numel(), ' rows:', int(d.any(1).sum()), ' columns:', d.any(0).tolist())
    r = d.any(1).nonzero()[:8, 0].tolist()
    for i in r: print('  row', i, 'parent', i // 8, 'child', i % 8, t[i].tolist(), t2[i].tolist())
for _ in range(200): pass_a(); pass_b()
torch.cuda.synchronize()
for rnd in range(3):
    print('  '.join(f'{nm} {med(f)[0]:7.1f} us (min {med(f)[1]:6.1f})' for nm, f in (('packedA', pass_a), ('quadA', pass_q), ('passB', pass_b), ('packedCls', cls_p), ('quadCls', cls_q))), flush=True)

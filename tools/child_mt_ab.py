#!/usr/bin/env python3
"""Two M tiles per wave (round 4) against the one-tile kernels: per-pass times of the children-level InceptionResNet passes, the plain
conv and the classification head on the stride-1 (C = 16) candidates of shell10, for the tuning codes given on the command line
(pcgc_set_child_tuning: 0 = product default, 200 + waves = two M tiles with ring depth 4, 300 + waves = ring depth 2), each checked
bit for bit against the default kernels' output."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd._lib import lib
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
dev = torch.device('cuda:0')
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16
codes = [int(a) for a in sys.argv[2:]] or [0, 208, 212, 308, 312, 316]
pts = synthetic.shell('shell10', device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
parent = CoordMap(c4, 1, unique=True).build_pyramid({16: 1, 32: 2, 64: 3}[C])
pk = parent.k3
n_p = len(parent); n = 8 * n_p
g = torch.Generator(device='cpu').manual_seed(0)
x = torch.randn((n, C), generator=g).to(dev)
blk = InceptionResNet(C).to(dev)
params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
with torch.no_grad():
    for p_ in params: p_.normal_(0, 0.1)
tabs = ops.child_irn_tables(params)
P = [p.data_ptr() for p in params]
W = (torch.randn((27, C, C), generator=g) * 0.05).to(dev); b = torch.randn((1, C), generator=g).to(dev)
tab = ops.child_conv_table(W)
Wc = (torch.randn((27, C, 1), generator=g) * 0.05).to(dev); tc = ops.child_cls_table(Wc)
s = torch.cuda.current_stream().cuda_stream
t = torch.empty((n, C // 2), device=dev); out = torch.empty((n, C), device=dev)
def pass_a(): ops.check(lib().pcgc_irn_child_pass(pk.data_ptr(), n_p, C, 1, x.data_ptr(), C, tabs[0].data_ptr(), tabs[0].numel() * 4, P[1], P[5], None, None, 0, t.data_ptr(), C // 2, s), 'a')
def pass_b(): ops.check(lib().pcgc_irn_child_pass(pk.data_ptr(), n_p, C, 2, t.data_ptr(), C // 2, tabs[1].data_ptr(), tabs[1].numel() * 4, P[3], P[7], P[9], x.data_ptr(), C, out.data_ptr(), C, s), 'b')
yc = [None]; ycls = [None]
def conv(): yc[0] = ops.conv_child(pk, x, tab, b, C, relu=True)
def cls(): ycls[0] = ops.conv_child(pk, x, tc, b[:, :1].contiguous(), 1)
def med(f, reps=15):
    for _ in range(3): f()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
print(f'{n} rows ({n_p} parents), C = {C}')
ref = {}
for _ in range(300): pass_a(); pass_b()                      # ~80 ms of work first: clocks and caches in their steady state
torch.cuda.synchronize()
for code in codes:
    ops.set_child_tuning(code, 0)
    row = []
    for name, f, res in (('passA', pass_a, lambda: t), ('passB', pass_b, lambda: out), ('conv', conv, lambda: yc[0]), ('cls', cls, lambda: ycls[0])):
        try:
            us = med(f)
        except Exception as e:
            row.append(f'{name} FAILED ({str(e)[:60]})'); continue
        r = res().clone()
        if code == codes[0]: ref[name] = r
        row.append(f'{name} {us:7.1f} us {"==" if torch.equal(r, ref[name]) else "DIFFERS"}')
    print(f'code {code:4d}: ' + '   '.join(row), flush=True)
ops.set_child_tuning(0, 0)

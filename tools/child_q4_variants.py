#!/usr/bin/env python3
"""Times experiment builds of the quad-block pass A (tools/experiments/q4/build_variants.sh -> tools/ubench/_bin/libq4_<name>.so) on the
stride-1 candidates of a cloud; `base` is also compared with the product library's packed-N pass A bit for bit."""
import os, sys, statistics, ctypes, glob
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd._lib import lib
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
dev = torch.device('cuda:0')
C = 16
cloud = sys.argv[1] if len(sys.argv) > 1 else 'shell10'
names = sys.argv[2:] or sorted(os.path.basename(p)[6:-3] for p in glob.glob(os.path.join(R, 'tools/ubench/_bin/libq4_*.so')))
pts = synthetic.shell(cloud, device=dev) if cloud in synthetic.SHELLS else synthetic.cloud(cloud, device=dev)
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
tabs = ops.child_irn_tables(params); tq = ops.child_q4_tables(params)
P = [p.data_ptr() for p in params]
s = torch.cuda.current_stream().cuda_stream
t = torch.empty((n, C // 2), device=dev); t2 = torch.full((n, C // 2), -7.0, device=dev)
out = torch.empty((n, C), device=dev); out2 = torch.full((n, C), -7.0, device=dev)
vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
def pass_a(): ops.check(lib().pcgc_irn_child_pass(pk.data_ptr(), n_p, C, 1, x.data_ptr(), C, tabs[0].data_ptr(), tabs[0].numel() * 4, P[1], P[5], None, None, 0, t.data_ptr(), C // 2, s), 'a')
def pass_b(): ops.check(lib().pcgc_irn_child_pass(pk.data_ptr(), n_p, C, 2, t.data_ptr(), C // 2, tabs[1].data_ptr(), tabs[1].numel() * 4, P[3], P[7], P[9], x.data_ptr(), C, out.data_ptr(), C, s), 'b')
def med(f, reps=15):
    for _ in range(3): f()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts), min(ts)
print(f'{cloud}: {n} rows ({n_p} parents)')
for _ in range(200): pass_a(); pass_b()
torch.cuda.synchronize()
print(f'packed-N (product)          pass A {med(pass_a)[0]:7.1f} us   pass B {med(pass_b)[0]:7.1f} us')
Wc = (torch.randn((27, C, 1), generator=g) * 0.05).to(dev); bc = torch.randn((1, 1), generator=g).to(dev)
tcq = ops.child_q4_cls_table(Wc); oc = torch.empty((n, 1), device=dev)
ref_cls = ops.conv_child(pk, x, ops.child_cls_table(Wc), bc, 1)
for nm in names:
    L = ctypes.CDLL(os.path.join(R, f'tools/ubench/_bin/libq4_{nm}.so'))
    fn = L.pcgc_irn_child_q4
    fn.restype, fn.argtypes = ci, [vp, i64, ci, ci, vp, ci, vp, i64, vp, vp, vp, vp, ci, vp, ci, vp]
    def pass_q(): 
        rc = fn(pk.data_ptr(), n_p, C, 1, x.data_ptr(), C, tq.data_ptr(), tq.numel() * 4, P[1], P[5], None, None, 0, t2.data_ptr(), C // 2, s)
        assert rc == 0, rc
    def pass_qb(tab=tabs[1]):
        rc = fn(pk.data_ptr(), n_p, C, 2, t2.data_ptr(), C // 2, tab.data_ptr(), tab.numel() * 4, P[3], P[7], P[9], x.data_ptr(), C, out2.data_ptr(), C, s)
        assert rc == 0, rc
    fc = L.pcgc_cls_child_q4; fc.restype, fc.argtypes = ci, [vp, i64, vp, ci, ci, vp, i64, vp, vp, vp]
    def cls_q():
        rc = fc(pk.data_ptr(), n_p, x.data_ptr(), C, C, tcq.data_ptr(), tcq.numel() * 4, bc.data_ptr(), oc.data_ptr(), s)
        assert rc == 0, rc
    cls_q()
    t2.fill_(-7.0); out2.fill_(-7.0); pass_q(); pass_qb(); torch.cuda.synchronize()
    m, lo = med(pass_q); mb, lob = med(pass_qb)
    mc, _ = med(cls_q)
    print(f'quad-block {nm:16s} pass A {m:7.1f} us (min {lo:6.1f})   pass B (T2 gather) {mb:7.1f} us   block == packed-N: {torch.equal(out, out2)}   cls {mc:7.1f} us == {torch.equal(ref_cls, oc)}', flush=True)

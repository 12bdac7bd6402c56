import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch, numpy as np
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd.sparse import CoordMap
dev = torch.device('cuda:0'); C = 16
def med(f, reps=15):
    for _ in range(3): f()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
for name in sys.argv[1:] or ['shell6', 'shell7', 'shell10']:
    pts = synthetic.shell(name, device=dev) if name in synthetic.SHELLS else synthetic.cloud(name, device=dev)
    c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
    parent = CoordMap(c4, 1, unique=True).build_pyramid(1)
    pk = parent.k3; n_p = len(parent); n = 8 * n_p
    g = torch.Generator(device='cpu').manual_seed(0)
    x = torch.randn((n, C), generator=g).to(dev)
    W = (torch.randn((27, C, 1), generator=g) * 0.05).to(dev); b = torch.randn((1, 1), generator=g).to(dev)
    tc = ops.child_cls_table(W); tq = ops.child_q4_cls_table(W)
    a = ops.conv_child(pk, x, tc, b, 1)
    q = ops.cls_child_q4(pk, x, tq, b)
    torch.cuda.synchronize()
    print(name, n, 'rows: quad-block cls == packed cls:', torch.equal(a, q), ' mismatches', int((a != q).sum()))
    if n > 1000000:
        for _ in range(100): ops.conv_child(pk, x, tc, b, 1)
        print('   packed %.1f us   quad-block %.1f us' % (med(lambda: ops.conv_child(pk, x, tc, b, 1)), med(lambda: ops.cls_child_q4(pk, x, tq, b))))

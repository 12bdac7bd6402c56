import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch, numpy as np
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd._lib import lib
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
dev = torch.device('cuda:0'); C = 16
for name in sys.argv[1:] or ['shell7', 'shell6', 'shell8']:
    pts = synthetic.shell(name, device=dev)
    c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
    parent = CoordMap(c4, 1, unique=True).build_pyramid(1)
    pk = parent.k3; n_p = len(parent); n = 8 * n_p
    g = torch.Generator(device='cpu').manual_seed(0)
    x = torch.randn((n, C), generator=g).to(dev)
    blk = InceptionResNet(C).to(dev)
    params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
    with torch.no_grad():
        for p_ in params: p_.normal_(0, 0.1)
    tabs = ops.child_irn_tables(params); tq = ops.child_q4_tables(params)
    P = [p.data_ptr() for p in params]
    s = torch.cuda.current_stream().cuda_stream
    t = torch.full((n, 8), -7.0, device=dev); t2 = torch.full((n, 8), -7.0, device=dev)
    ops.check(lib().pcgc_irn_child_pass(pk.data_ptr(), n_p, C, 1, x.data_ptr(), C, tabs[0].data_ptr(), tabs[0].numel() * 4, P[1], P[5], None, None, 0, t.data_ptr(), 8, s), 'a')
    ops.check(lib().pcgc_irn_child_q4(pk.data_ptr(), n_p, C, 1, x.data_ptr(), C, tq.data_ptr(), tq.numel() * 4, P[1], P[5], None, None, 0, t2.data_ptr(), 8, s), 'q')
    torch.cuda.synchronize()
    std = t2.view(n_p, 2, 2, 4, 4).permute(0, 1, 3, 2, 4).reshape(n, 8)
    d = (std != t)
    print(name, 'parents', n_p, 'tiles', (n_p + 127) // 128, 'pass A mismatching elements', int(d.sum()), 'of', d.numel(), 'untouched(-7):', int((t2 == -7).sum()))
    if d.any():
        rows = d.any(1).nonzero()[:, 0]
        par = (rows // 8).unique()
        print('  mismatching parents:', len(par), 'first', par[:10].tolist(), 'last', par[-5:].tolist(), ' by tile:', torch.bincount(par // 128).tolist())
        print('  by child:', torch.bincount(rows % 8, minlength=8).tolist(), ' by column:', d.sum(0).tolist())
    # pass B in isolation: the packed pass A's t converted to the T2 layout -> the T2 pass B == the standard pass B on t
    out = torch.full((n, C), -7.0, device=dev); out2 = torch.full((n, C), -7.0, device=dev); out3 = torch.full((n, C), -7.0, device=dev)
    t_as_t2 = t.view(n_p, 2, 4, 2, 4).permute(0, 1, 3, 2, 4).contiguous().view(n, 8)
    ops.check(lib().pcgc_irn_child_pass(pk.data_ptr(), n_p, C, 2, t.data_ptr(), 8, tabs[1].data_ptr(), tabs[1].numel() * 4, P[3], P[7], P[9], x.data_ptr(), C, out.data_ptr(), C, s), 'b')
    ops.check(lib().pcgc_irn_child_q4(pk.data_ptr(), n_p, C, 2, t_as_t2.data_ptr(), 8, tabs[1].data_ptr(), tabs[1].numel() * 4, P[3], P[7], P[9], x.data_ptr(), C, out2.data_ptr(), C, s), 'qb')
    ops.check(lib().pcgc_irn_child_q4(pk.data_ptr(), n_p, C, 2, t2.data_ptr(), 8, tabs[1].data_ptr(), tabs[1].numel() * 4, P[3], P[7], P[9], x.data_ptr(), C, out3.data_ptr(), C, s), 'qb')
    torch.cuda.synchronize()
    print('   pass B (T2 of converted t) == standard:', torch.equal(out, out2), '  (of q4 t):', torch.equal(out, out3), ' t2 == converted t:', torch.equal(t2, t_as_t2))
    got = ops.irn_block_child(pk, x, params, tabs, q4_table=tq)
    print('   ops.irn_block_child(q4) == standard:', torch.equal(out, got))

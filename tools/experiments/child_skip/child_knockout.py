"""Where do the children-level kernels spend their time?  Knock-out timing of ONE kernel (pass A / pass B of an InceptionResNet, or a
plain conv) on the stride-1 children level of shell10's true geometry: full kernel vs no gather memory access vs no MFMAs vs no LDS
operand reads (results are wrong in the knock-outs; timing only).    python tools/child_knockout.py [C] [sorted_chunk]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd._lib import lib
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
DEV = torch.device('cuda:0')
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
name = {16: 'shell10', 32: 'shell10', 64: 'shell10'}[C]
p = synthetic.shell(name, device=DEV)
c4 = torch.cat([torch.zeros((len(p), 1), dtype=torch.int32, device=DEV), p], 1).contiguous()
lvl = CoordMap(c4, 1, unique=True)
for _ in range({16: 1, 32: 2, 64: 3}[C]):
    lvl = lvl.down()[0]
n = 8 * len(lvl)
blk = InceptionResNet(C).to(DEV)
params = [q for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for q in (m.kernel, m.bias)]
tables = ops.child_irn_tables(params)
x = torch.randn((n, C), device=DEV)
W = torch.randn((27, C, C), device=DEV) * 0.05
tab = ops.child_conv_table(W) if C in (16, 32) else None
bias = torch.zeros((1, C), device=DEV)
def timeit(f, reps=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print(f'C={C} parents {len(lvl)} rows {n}')
for label, sort in (('canonical', False), (f'sorted chunk {chunk}', True)):
    ops.CHILD_SORT_CHUNK = chunk
    g = ops.ChildGeom(lvl.k3, sort=sort)
    print(f'--- {label}: live cell fraction {g.live_fraction():.3f}')
    irn = ops.irn_block_child64 if C == 64 else ops.irn_block_child
    for mode, tag in ((0, 'no skip (every cell)'), (1, 'skip'), (-2, 'skip, gathers touch no memory'), (-4, 'skip, no MFMAs'), (-8, 'skip, no LDS operand reads'),
                      (-6, 'skip, no memory, no MFMAs'), (-14, 'skip, nothing but the control flow + DMA issue + epilogue'), (-30, '... and no epilogue traffic'), (-62, '... and no DMA issue: prologue + branches only'), (-16, 'skip, full main loop, no epilogue traffic')):
        ops.set_child_skip(mode)
        t_irn = timeit(lambda: irn(g, x, params, tables))
        t_conv = timeit(lambda: ops.conv_child(g, x, tab, bias, C)) if tab is not None else float('nan')
        print(f'  {tag:58s} IRN block (A+B) {t_irn:7.1f} us   conv {C}->{C} {t_conv:7.1f} us')
ops.set_child_skip(1)

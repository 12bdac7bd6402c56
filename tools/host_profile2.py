#!/usr/bin/env python3
"""cProfile of the host side of encode+decode on a SMALL cloud (host-bound regime: the blocks of config 5)."""
import os, sys, tempfile, cProfile, pstats, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pcgcv2_amd
pcgcv2_amd.configure_host_threads()
from pcgcv2_amd import synthetic
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'shell9'
p = synthetic.shell(name, device=dev)
c = torch.cat([torch.zeros((len(p), 1), dtype=torch.int32, device=dev), p], 1).contiguous()
x = SparseTensor(torch.ones((len(p), 1), device=dev), coordinates=c, tensor_stride=1, device=dev)
m = PCCModel().to(dev); m.load_state_dict(synthetic.synthetic_state_dict())
coder = Coder(m, os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'f'))
def step():
    x.cmap.drop_caches(); coder.encode(x); coder.decode(); torch.cuda.synchronize()
for _ in range(5): step()
t = time.perf_counter()
for _ in range(20): step()
print(f'{name}: {len(p)} points, {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per encode+decode')
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)

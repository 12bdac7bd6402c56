#!/usr/bin/env python3
"""cProfile of the host side of one warmed-up encode + decode (where the interpreter's time goes between launches)."""
import cProfile, os, pstats, sys, tempfile, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcgcv2_amd
pcgcv2_amd.configure_host_threads()
from pcgcv2_amd import synthetic
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
coords = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
x = SparseTensor(torch.ones((len(pts), 1), device=dev), coordinates=coords, tensor_stride=1, device=dev)
model = PCCModel().to(dev); model.load_state_dict(synthetic.synthetic_state_dict())
coder = Coder(model, os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'f'))
def step():
    x.cmap.drop_caches(); coder.encode(x); out = coder.decode(); torch.cuda.synchronize(); return out
for _ in range(5): step()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])

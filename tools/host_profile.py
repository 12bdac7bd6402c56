#!/usr/bin/env python3
"""cProfile of one warmed-up encode+decode step (host-side view)."""
import cProfile, pstats, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import synthetic
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor

dev = torch.device('cuda:0')
pts = synthetic.shell(sys.argv[1] if len(sys.argv) > 1 else 'shell10', device=dev)
coords = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
feats = torch.ones((len(pts), 1), device=dev)
model = PCCModel().to(dev); model.load_state_dict(synthetic.synthetic_state_dict())
coder = Coder(model, os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'f'))

def step():
    x = SparseTensor(feats, coordinates=coords, tensor_stride=1, device=dev)
    coder.encode(x); out = coder.decode(); torch.cuda.synchronize(); return out

for _ in range(3): step()
t = time.perf_counter(); step(); print('step ms', (time.perf_counter() - t) * 1e3)
pr = cProfile.Profile(); pr.enable(); step(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)

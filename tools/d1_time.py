#!/usr/bin/env python3
"""Time of the device D1 metric on the bench frame (input cloud vs the cloud the random-weight stand-in decodes): cells vs lattice-offset probes."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcgcv2_amd
pcgcv2_amd.configure_host_threads()
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor
from pcgcv2_amd.pc_error import d1_psnr_device
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
coords = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
model = PCCModel().to(dev); model.load_state_dict(synthetic.synthetic_state_dict())
coder = Coder(model, os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'f'))
x = SparseTensor(torch.ones((len(pts), 1), device=dev), coordinates=coords, tensor_stride=1, device=dev)
coder.encode(x); out = coder.decode()
res = {}
for cells in (True, False, True):
    ops.D1_CELLS = cells
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); m = d1_psnr_device(x.C, out.C, 1024); dt = time.perf_counter() - t
    res[cells] = m
    print('cells' if cells else 'probes', f'{dt * 1e3:.2f} ms', 'mseF PSNR', round(m['mseF,PSNR (p2point)'], 4), 'h.', m['h.        (p2point)'])
print('identical', res[True] == res[False])

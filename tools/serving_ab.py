#!/usr/bin/env python3
"""Serving throughput (4 frames in flight, shard.code_units) for different sizes of the entropy-decoder pools, five repetitions each."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcgcv2_amd
pcgcv2_amd.configure_host_threads()
from pcgcv2_amd import synthetic, ops, shard
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor
dev = torch.device('cuda:0')
model = PCCModel().to(dev); model.load_state_dict(synthetic.synthetic_state_dict())
tmp = tempfile.mkdtemp(dir='/dev/shm')
units = []
for i in range(16):
    pts = synthetic.shell(['shell10', 'shell10_b', 'shell10_c', 'shell10_d'][i % 4], device=dev)
    coords = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
    units.append((f's{i}', SparseTensor(torch.ones((len(pts), 1), device=dev), coordinates=coords, tensor_stride=1, device=dev)))
coder = Coder(model, os.path.join(tmp, 'f'))
n = sum(len(u) for _, u in units)
def run():
    for _, u in units: u.cmap.drop_caches()
    t = time.perf_counter(); shard.code_units(coder, units, in_flight=4); torch.cuda.synchronize()
    return n / (time.perf_counter() - t) / 1e6
run()
for rc_threads in (8, 3, 1):
    ops.set_rc_threads(rc_threads)
    vals = [run() for _ in range(5)]
    print(f'entropy pools of {rc_threads} thread(s): ' + ' '.join(f'{v:6.1f}' for v in vals) + f'   median {sorted(vals)[2]:.1f} Mpoints/s')

#!/usr/bin/env python3
"""Wall-clock attribution of one encode+decode (device-synchronised after every phase; medians over N runs)."""
import os, sys, tempfile, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcgcv2_amd
pcgcv2_amd.configure_host_threads()
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor
from pcgcv2_amd.data_utils import sort_spare_tensor

dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
coords = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
feats = torch.ones((len(pts), 1), device=dev)
model = PCCModel().to(dev); model.load_state_dict(synthetic.synthetic_state_dict())
coder = Coder(model, os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'f'))
T = {}
def lap(name, t0):
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3; T.setdefault(name, []).append(dt); return time.perf_counter()
for it in range(14):
    t = time.perf_counter()
    x = SparseTensor(feats, coordinates=coords, tensor_stride=1, device=dev); t = lap('enc.0 tensor+dedup', t)
    with torch.no_grad():
        ys = model.encoder(x); t = lap('enc.1 encoder net', t)
        y = sort_spare_tensor(ys[0]); t = lap('enc.2 sort', t)
        coder.feature_coder.encode(y.F); t = lap('enc.3 feature coder (tables+rc+files)', t)
        coder.coordinate_coder.encode((y.C // 8).cpu()[:, 1:]); t = lap('enc.4 coordinate coder', t)
        yc = coder.coordinate_coder.decode(); t = lap('dec.0 coordinate decode', t)
        yc = torch.cat((torch.zeros((len(yc), 1)).int(), torch.tensor(yc).int()), dim=-1); yc = (yc * 8).to(dev)
        yc = ops.gather_coords(yc, ops.sort_zyx(yc)); t = lap('dec.1 H2D+sort', t)
        yf = coder.feature_coder.decode(device=dev); t = lap('dec.2 feature decode (tables+rc)', t)
        yy = SparseTensor(yf, coordinates=yc, tensor_stride=8, device=dev, assume_unique=True)
        nums = [[len(ys[1])], [len(ys[2])], [len(x)]]
        _, out = model.decoder(yy, nums); t = lap('dec.3 decoder net', t)
print('per-iteration (ms):')
for k, v in T.items():
    print(f'{k[:28]:28s}', ' '.join(f'{x:6.1f}' for x in v))
tot = 0
for k, v in T.items():
    m = statistics.median(v[2:]); tot += m
    print(f'{k:45s} {m:8.3f} ms')
print(f'{"sum of medians":45s} {tot:8.3f} ms')

#!/usr/bin/env python3
"""Soak: N encode+decode cycles of shell10 (alternating with shell9 so that buffer sizes change); host RSS and device memory at step 100 and at the end."""
import os, sys, tempfile, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcgcv2_amd
pcgcv2_amd.configure_host_threads()
from pcgcv2_amd import synthetic
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device('cuda:0')
model = PCCModel().to(dev); model.load_state_dict(synthetic.synthetic_state_dict())
coder = Coder(model, os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'f'))
clouds = []
from pcgcv2_amd import entropy_model
for name, order in (('shell10', 'raster'), ('noisy_s', 'shuffled')):      # (round 4: the second cloud is unordered — ingest sort — and not a shell)
    pts = synthetic.cloud(name, order=order, seed=2).to(dev)
    c = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
    clouds.append(SparseTensor(torch.ones((len(pts), 1), device=dev), coordinates=c, tensor_stride=1, device=dev))
def rss_mb():
    return int(open('/proc/self/statm').read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 2 ** 20
ref = {}
t0 = time.perf_counter()
for i in range(N):
    x = clouds[i % 7 == 6]
    x.cmap.drop_caches()
    if i % 3 == 0: entropy_model.table_cache(clear=True)      # cold and warm tables alternate
    coder.encode(x); out = coder.decode()
    n = len(out)
    key = len(x)
    if key not in ref: ref[key] = out.C.clone()
    elif i % 50 == 0: assert torch.equal(out.C, ref[key]), f'step {i}: decoded cloud changed'
    if i in (100, N - 1):
        torch.cuda.synchronize()
        print(f'step {i}: RSS {rss_mb():.0f} MB, device allocated {torch.cuda.memory_allocated() / 2**20:.0f} MB, reserved {torch.cuda.memory_reserved() / 2**20:.0f} MB, threads {len(os.listdir("/proc/self/task"))}')
torch.cuda.synchronize()
print(f'{N} cycles in {time.perf_counter() - t0:.1f} s')

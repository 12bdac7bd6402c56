#!/usr/bin/env python3
"""Throughput with F frames in flight per GPU: F host threads, each with its own Coder, HIP stream and file prefix, code
independent frames concurrently (serving mode).  Prints Mpoints/s for F = 1, 2, 3, 4.  The headline bench keeps F = 1."""
import os, sys, tempfile, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcgcv2_amd
pcgcv2_amd.configure_host_threads()
from pcgcv2_amd import synthetic
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor
import pcgcv2_amd.coder as coder_mod
from concurrent.futures import ThreadPoolExecutor

dev = torch.device('cuda:0')
model = PCCModel().to(dev); model.load_state_dict(synthetic.synthetic_state_dict())
tmp = tempfile.mkdtemp(dir='/dev/shm')
names = ['shell10', 'shell10_b', 'shell10_c', 'shell10_d']
frames = []
for nm in names:
    pts = synthetic.shell(nm, device=dev)
    coords = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
    frames.append(SparseTensor(torch.ones((len(pts), 1), device=dev), coordinates=coords, tensor_stride=1, device=dev))
coder_mod._POOL = ThreadPoolExecutor(max_workers=8, thread_name_prefix='pcgc-coord')     # one helper per frame in flight

def worker(i, steps, barrier, out):
    torch.cuda.set_device(dev)
    stream = torch.cuda.Stream(device=dev)
    coder = Coder(model, os.path.join(tmp, f'f{i}'))
    x = frames[i]
    with torch.cuda.stream(stream):
        for s in range(steps + 2):
            if s == 2:
                stream.synchronize(); barrier.wait(); t0 = time.perf_counter()
            x.cmap.drop_caches()
            coder.encode(x)
            coder.decode()
        stream.synchronize()
    out[i] = (t0, time.perf_counter())

for F in (1, 2, 3, 4):
    steps = 10
    barrier = threading.Barrier(F); out = {}
    th = [threading.Thread(target=worker, args=(i, steps, barrier, out)) for i in range(F)]
    [t.start() for t in th]; [t.join() for t in th]
    t0 = min(v[0] for v in out.values()); t1 = max(v[1] for v in out.values())
    pts = sum(len(frames[i]) for i in range(F)) * steps
    print(f'F={F}: {pts / (t1 - t0) / 1e6:7.2f} Mpoints/s   ({(t1 - t0) / steps * 1e3:.2f} ms per round of {F} frame(s))')

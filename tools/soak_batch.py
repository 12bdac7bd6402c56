#!/usr/bin/env python3
"""Soak of the batch API: N encode_batch + decode_batch cycles of four collated clouds (alternating two batch compositions); host RSS, device
memory and thread count at cycle 50 and at the end; the decoded clouds must not change."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcgcv2_amd
pcgcv2_amd.configure_host_threads()
from pcgcv2_amd import synthetic
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda:0')
model = PCCModel().to(dev); model.load_state_dict(synthetic.synthetic_state_dict())
coder = Coder(model, os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'b'))
def collate(names):
    parts = []
    for b, n in enumerate(names):
        p = synthetic.shell(n, device=dev)
        parts.append(torch.cat([torch.full((len(p), 1), b, dtype=torch.int32, device=dev), p], 1))
    c = torch.cat(parts).contiguous()
    return SparseTensor(torch.ones((len(c), 1), device=dev), coordinates=c, tensor_stride=1, device=dev)
batches = [collate(['shell9', 'shell8', 'shell9', 'shell7']), collate(['shell8', 'shell9', 'shell7', 'shell8'])]
post = ['_0', '_1', '_2', '_3']
def rss_mb():
    return int(open('/proc/self/statm').read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 2 ** 20
ref = {}
t0 = time.perf_counter()
for i in range(N):
    k = i % 2
    x = batches[k]
    x.cmap.drop_caches()
    coder.encode_batch(x, post)
    outs = coder.decode_batch(post)
    if k not in ref: ref[k] = [o.C.clone() for o in outs]
    elif i % 20 < 2:
        assert all(torch.equal(o.C, r) for o, r in zip(outs, ref[k])), f'cycle {i}: decoded clouds changed'
    if i in (50, N - 1):
        torch.cuda.synchronize()
        print(f'cycle {i}: RSS {rss_mb():.0f} MB, device allocated {torch.cuda.memory_allocated() / 2**20:.0f} MB, reserved {torch.cuda.memory_reserved() / 2**20:.0f} MB, threads {len(os.listdir("/proc/self/task"))}')
print(f'{N} batch cycles in {time.perf_counter() - t0:.1f} s')

#!/usr/bin/env python3
"""Where do the ~120 small copies / fills per encode+decode come from?  torch.profiler over one step, grouped by call site."""
import os, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import pcgcv2_amd
pcgcv2_amd.configure_host_threads()
from pcgcv2_amd import synthetic
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor

dev = torch.device('cuda:0')
p = synthetic.shell('shell10', device=dev)
c = torch.cat([torch.zeros((len(p), 1), dtype=torch.int32, device=dev), p], 1).contiguous()
x = SparseTensor(torch.ones((len(p), 1), device=dev), coordinates=c, tensor_stride=1, device=dev)
m = PCCModel().to(dev); m.load_state_dict(synthetic.synthetic_state_dict())
coder = Coder(m, os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'f'))
for _ in range(3):
    x.cmap.drop_caches(); coder.encode(x); coder.decode()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    x.cmap.drop_caches(); coder.encode(x); coder.decode(); torch.cuda.synchronize()
ev = prof.events()
cnt = collections.Counter()
for e in ev:
    if e.name.startswith('aten::') and e.name in ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::item', 'aten::_local_scalar_dense', 'aten::cat',
                                                    'aten::empty', 'aten::to', 'aten::_to_copy', 'aten::contiguous', 'aten::clone', 'aten::zeros', 'aten::full',
                                                    'aten::sum', 'aten::index', 'aten::ge'):
        st = [s for s in (e.stack or []) if 'pcgcv2_amd' in s or 'coder' in s]
        cnt[(e.name, st[0].split('/')[-1] if st else '?')] += 1
for (name, site), n in cnt.most_common(60):
    print(f'{n:4d}  {name:28s} {site}')
gpu = collections.Counter(e.name for e in ev if e.device_type is not None and str(e.device_type).endswith('CUDA'))
print('--- device activities'); [print(f'{n:5d} {k[:90]}') for k, n in gpu.most_common(12)]

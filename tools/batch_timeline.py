#!/usr/bin/env python3
"""Host timeline of one warmed-up encode_batch + decode_batch of the 8 octant blocks of config 5 (wall-clock phases, no extra syncs)."""
import os, sys, tempfile, threading, time, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcgcv2_amd
pcgcv2_amd.configure_host_threads()
from pcgcv2_amd import synthetic, ops, coder as coder_mod, sparse, shard
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor
from pcgcv2_amd.data_utils import scale_sparse_tensor
LOG = []; T0 = [0.0]; DEPTH = threading.local()
def wrap(owner, name, label=None):
    fn = getattr(owner, name)
    @functools.wraps(fn)
    def w(*a, **k):
        d = getattr(DEPTH, 'v', 0); DEPTH.v = d + 1
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            DEPTH.v = d
            LOG.append((t - T0[0], time.perf_counter() - T0[0], d, threading.current_thread().name[:12], label or name))
    setattr(owner, name, w)
dev = torch.device('cuda:0')
p = synthetic.shell('shell12', device=dev)
c = torch.cat([torch.zeros((len(p), 1), dtype=torch.int32, device=dev), p], 1).contiguous()
whole = SparseTensor(torch.ones((len(p), 1), device=dev), coordinates=c, tensor_stride=1, device=dev)
x_in = scale_sparse_tensor(whole, 0.375)
blocks = shard.split_octants(x_in.C, levels=1)
cb = torch.cat([torch.cat([torch.full((len(b), 1), i, dtype=torch.int32, device=dev), x_in.C[torch.as_tensor(b, device=dev)][:, 1:]], 1) for i, b in enumerate(blocks)], 0).contiguous()
xb = SparseTensor(torch.ones((len(cb), 1), device=dev), coordinates=cb, tensor_stride=1, device=dev, assume_unique=True)
posts = [f'_b{i}' for i in range(len(blocks))]
model = PCCModel().to(dev); model.load_state_dict(synthetic.synthetic_state_dict())
coder = Coder(model, os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'f'))
for cls, names in ((Coder, ['encode_batch', 'decode_batch']), (coder_mod.FeatureCoder, ['encode_symbols', 'decode_symbols']),
                   (coder_mod.CoordinateCoder, ['encode', 'decode']), (type(model.encoder), ['forward']), (type(model.decoder), ['forward'])):
    for n in names: wrap(cls, n, f'{cls.__name__}.{n}')
for n in ('quantize_symbols_segments', 'sort_zyx', 'desymbolize', 'topk_mask_segments', 'batch_counts', 'pyramid', 'gather_coords', 'gather_feats'):
    wrap(ops, n, 'ops.' + n)
def step():
    xb.cmap.drop_caches(); coder.encode_batch(xb, posts); outs = coder.decode_batch(posts); torch.cuda.synchronize(); return outs
for _ in range(4): step()
LOG.clear(); torch.cuda.synchronize(); T0[0] = time.perf_counter()
step()
print(f'step {1e3 * (time.perf_counter() - T0[0]):.3f} ms')
for a, b, d, th, name in sorted(LOG):
    if th.startswith('pcgc-item') and d > 0: continue
    print(f'{1e3 * a:8.3f} -> {1e3 * b:8.3f}  ({1e3 * (b - a):6.3f})  {th:12s} {"  " * d}{name}')

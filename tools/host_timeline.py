#!/usr/bin/env python3
"""Host-side timeline of one warmed-up encode+decode: wall-clock start/end (ms from step start) of the main host phases on
both threads, WITHOUT extra device synchronisation (what the bench's step actually does)."""
import os, sys, tempfile, threading, time, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcgcv2_amd
pcgcv2_amd.configure_host_threads()
from pcgcv2_amd import synthetic, ops, coder as coder_mod, sparse
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor

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
            LOG.append((t - T0[0], time.perf_counter() - T0[0], d, threading.current_thread().name[:10], label or name))
    setattr(owner, name, w)

dev = torch.device('cuda:0')
pts = synthetic.shell(sys.argv[1] if len(sys.argv) > 1 else 'shell10', device=dev)
coords = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
feats = torch.ones((len(pts), 1), device=dev)
model = PCCModel().to(dev); model.load_state_dict(synthetic.synthetic_state_dict())
coder = Coder(model, os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'f'))
x = SparseTensor(feats, coordinates=coords, tensor_stride=1, device=dev)
for cls, names in ((Coder, ['encode', 'decode', '_decode_geometry', '_stage_geometry', '_sort_and_stage']), (coder_mod.FeatureCoder, ['encode', 'decode']),
                   (coder_mod.CoordinateCoder, ['encode', 'decode']), (sparse.CoordMap, ['down', 'prepare_up', 'build_pyramid']), (Coder, ['_ingest', '_stage_level', '_decode_buffers', '_upload_level']),
                   (type(model.encoder), ['forward']), (type(model.decoder), ['forward'])):
    for n in names:
        wrap(cls, n, f'{cls.__name__}.{n}')
for n in ('rc_encode', 'rc_decode', 'quantize_symbols', 'sort_zyx', 'desymbolize', 'topk_mask', 'items_encode', 'items_probe', 'items_decode', 'frame_decode', 'frame_decode_begin', 'frame_decode_end', 'table_warm', 'pyramid', 'level_prepare_children', 'conv_up2', 'gather_feats'):
    wrap(ops, n, 'ops.' + n)
wrap(coder_mod, '_dump'); wrap(coder_mod, '_slurp')
from pcgcv2_amd import entropy_model
wrap(entropy_model.EntropyBottleneck, 'host_table', 'EB.host_table')

def step():
    x.cmap.drop_caches()
    entropy_model.table_cache(clear=True)
    coder.encode(x); entropy_model.table_cache(clear=True)
    out = coder.decode(); torch.cuda.synchronize(); return out
for _ in range(4): step()
LOG.clear(); torch.cuda.synchronize(); T0[0] = time.perf_counter()
step()
print(f'step {1e3 * (time.perf_counter() - T0[0]):.3f} ms')
for a, b, d, th, name in sorted(LOG):
    print(f'{1e3 * a:8.3f} -> {1e3 * b:8.3f}  ({1e3 * (b - a):6.3f})  {th:10s} {"  " * d}{name}')

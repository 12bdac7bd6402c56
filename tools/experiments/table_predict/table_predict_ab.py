import os, sys, time, tempfile
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import pcgcv2_amd
pcgcv2_amd.configure_host_threads(local_world=1)
from pcgcv2_amd import synthetic, ops, entropy_model, coder as coder_mod
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor
dev = torch.device('cuda:0')
sd = synthetic.synthetic_state_dict(); model = PCCModel().to(dev); model.load_state_dict(sd)
p = synthetic.shell('shell10', device=dev)
c = torch.cat([torch.zeros((len(p), 1), dtype=torch.int32, device=dev), p], 1).contiguous()
x = SparseTensor(torch.ones((len(p), 1), device=dev), coordinates=c, tensor_stride=1, device=dev)
tmp = tempfile.mkdtemp(dir='/dev/shm')
coder = Coder(model, os.path.join(tmp, 'u'))
coder.encode(x); coder.decode(); torch.cuda.synchronize()
em = coder.feature_coder.entropy_model
g = coder._range_guess
print('range', g, 'torch threads', torch.get_num_threads())
P = em._host_packed()
for label, fn in (('table_warm', lambda: ops.table_warm(P, em._channels)), ('prefetch cold', lambda: ops.table_prefetch(P, em._channels, *g))):
    ts = []
    for _ in range(8):
        entropy_model.table_cache(clear=True)
        t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3)
    print(label, ' '.join(f'{v:.3f}' for v in ts))
ts = []
for _ in range(5):
    t = time.perf_counter(); ops.table_prefetch(P, em._channels, *g); ts.append((time.perf_counter() - t) * 1e3)
print('prefetch cached', ' '.join(f'{v:.3f}' for v in ts))
def run(pred, n=15):
    coder_mod.PREDICT_TABLE_RANGE = pred
    for _ in range(3):
        x.cmap.drop_caches(); entropy_model.table_cache(clear=True); coder.encode(x); entropy_model.table_cache(clear=True); coder.decode()
    torch.cuda.synchronize(); e = d = 0.0
    for _ in range(n):
        x.cmap.drop_caches()
        a = time.perf_counter(); entropy_model.table_cache(clear=True); coder.encode(x); torch.cuda.synchronize()
        b = time.perf_counter(); entropy_model.table_cache(clear=True); coder.decode(); torch.cuda.synchronize()
        cc = time.perf_counter(); e += b - a; d += cc - b
    return e / n * 1e3, d / n * 1e3
for rep in range(3):
    for pred in (False, True):
        e, d = run(pred)
        print(f'predict={pred}: enc {e:.3f} dec {d:.3f} total {e + d:.3f} ms')

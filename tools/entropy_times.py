#!/usr/bin/env python3
"""Wall-clock attribution inside FeatureCoder.encode / decode for the shell10 latent (medians, device-synchronised)."""
import os, sys, tempfile, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcgcv2_amd
pcgcv2_amd.configure_host_threads()
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor

dev = torch.device('cuda:0')
pts = synthetic.shell(sys.argv[1] if len(sys.argv) > 1 else 'shell10', device=dev)
coords = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
feats = torch.ones((len(pts), 1), device=dev)
model = PCCModel().to(dev); model.load_state_dict(synthetic.synthetic_state_dict())
prefix = os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'f')
coder = Coder(model, prefix)
x = SparseTensor(feats, coordinates=coords, tensor_stride=1, device=dev)
with torch.no_grad():
    y = coder.encode(x)
eb = model.entropy_bottleneck
F = y.F.contiguous()
T = {}
def lap(name, t0, sync=True):
    if sync: torch.cuda.synchronize()
    T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3); return time.perf_counter()
for it in range(30):
    torch.cuda.synchronize(); t = time.perf_counter()
    prep = ops.compress_prepare(F, eb.packed_params(dev), 8); t = lap('enc: compress_prepare (kernels + D2H)', t)
    min_v, max_v, sym_h, tab = prep
    s = ops.rc_encode(tab, sym_h); t = lap('enc: rc_encode', t, False)
    open(prefix + '_F.bin', 'wb').write(s); t = lap('enc: write _F.bin', t, False)
    payload = open(prefix + '_F.bin', 'rb').read(); t = lap('dec: read _F.bin', t, False)
    table, _ = eb.cdf_table(min_v, max_v, dev); th = table.cpu().numpy().view(np.uint16); t = lap('dec: cdf_table kernel + D2H', t)
    sh = ops.rc_decode(th, payload, F.numel()); t = lap('dec: rc_decode', t, False)
    sym = torch.from_numpy(sh.reshape(F.shape)).to(dev); out = ops.desymbolize(sym, min_v); t = lap('dec: H2D + desymbolize', t)
    assert np.array_equal(sh.reshape(-1), sym_h.reshape(-1))
print(f'symbols {F.numel()}  bytes {len(payload)}  L {tab.shape[1] - 1}')
for k, v in T.items():
    print(f'{k:45s} {statistics.median(v[5:]):8.3f} ms')

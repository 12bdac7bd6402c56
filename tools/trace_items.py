import os, sys, tempfile, time
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/pcgcv2_amd') else '.')
import numpy as np, torch
import pcgcv2_amd
pcgcv2_amd.configure_host_threads()
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
coords = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
model = PCCModel().to(dev); model.load_state_dict(synthetic.synthetic_state_dict())
coder = Coder(model, os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'f'))
x = SparseTensor(torch.ones((len(pts), 1), device=dev), coordinates=coords, tensor_stride=1, device=dev)
for i in range(8):
    x.cmap.drop_caches()
    sys.stderr.write(f'--- step {i}\n')
    coder.encode(x); coder.decode(); torch.cuda.synchronize()

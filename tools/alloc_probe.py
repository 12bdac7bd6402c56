import os, sys, tempfile, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from pcgcv2_amd import synthetic
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
coords = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
feats = torch.ones((len(pts), 1), device=dev)
model = PCCModel().to(dev); model.load_state_dict(synthetic.synthetic_state_dict())
coder = Coder(model, os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'f'))
for it in range(12):
    torch.cuda.synchronize(); t = time.perf_counter()
    x = SparseTensor(feats, coordinates=coords, tensor_stride=1, device=dev)
    coder.encode(x); out = coder.decode(); torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    print(it, f'{(time.perf_counter()-t)*1e3:6.1f} ms', 'device_alloc', st.get('num_device_alloc'), 'device_free', st.get('num_device_free'),
          'retries', st.get('num_alloc_retries'), 'reserved MB', st['reserved_bytes.all.current'] >> 20, 'peak alloc MB', st['allocated_bytes.all.peak'] >> 20)

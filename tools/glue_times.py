import os, sys, time, tempfile
sys.path.insert(0, '/root/repo')
import torch
import pcgcv2_amd
pcgcv2_amd.configure_host_threads()
from pcgcv2_amd import synthetic, gpcc, entropy_model, coder as cm
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
dev = torch.device('cuda:0')
model = PCCModel().to(dev); model.load_state_dict(synthetic.synthetic_state_dict())
coder = Coder(model, os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'f'))
eb = model.entropy_bottleneck
def t(name, f, n=2000):
    f(); a = time.perf_counter()
    for _ in range(n): f()
    print(f'{name:40s} {1e6 * (time.perf_counter() - a) / n:7.2f} us')
t('next(decoder.parameters()).device', lambda: next(model.decoder.parameters()).device)
t('gpcc.tmc3_path()', gpcc.tmc3_path)
t('coder._native_items()', coder._native_items)
t('eb._stamp()', eb._stamp)
t('eb._host_packed()', eb._host_packed)
t('torch.cuda.current_stream(dev)', lambda: torch.cuda.current_stream(dev))
def w():
    with torch.cuda.device(dev): pass
t('with torch.cuda.device(dev)', w)
def ng():
    with torch.no_grad(): pass
t('with torch.no_grad()', ng)
t('entropy_model.table_cache(clear=True)', lambda: entropy_model.table_cache(clear=True))
t('coder._decode_buffers(8)', lambda: coder._decode_buffers(8))
t('torch.cuda.synchronize()', torch.cuda.synchronize)
ev = torch.cuda.Event(); ev.record()
t('event.synchronize()', ev.synchronize)

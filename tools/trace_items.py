#!/usr/bin/env python3
"""Per-stage microseconds of the native item codec inside real frames (PCGC_ITEMS_TRACE): medians over N encode + decode steps of shell10.
RC_THREADS=n sets the segment / group thread count (default: the library's choice)."""
import os, re, sys, tempfile, statistics, collections
os.environ['PCGC_ITEMS_TRACE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
if os.environ.get('RC_THREADS'):
    ops.set_rc_threads(int(os.environ['RC_THREADS']))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
log = tempfile.NamedTemporaryFile(prefix='items_trace_', suffix='.log', delete=False)
sys.stderr.flush()
keep = os.dup(2)
os.dup2(log.fileno(), 2)
try:
    from pcgcv2_amd import entropy_model, coder as coder_mod
    cold = os.environ.get('COLD_TABLES', '1') != '0'               # bench.py's default: every encode and every decode evaluates its table
    coder_mod.WARM_TABLE_CODE = os.environ.get('WARM_CODE', '1') != '0'
    for i in range(N + 5):
        x.cmap.drop_caches()
        if cold: entropy_model.table_cache(clear=True)
        coder.encode(x)
        if cold: entropy_model.table_cache(clear=True)
        coder.decode(); torch.cuda.synchronize()
finally:
    os.dup2(keep, 2)
stages = collections.defaultdict(list)
for line in open(log.name).read().splitlines()[5 * 3:]:
    m = re.match(r'\[pcgc items\] (.*?) \d+:(.*)\| total (\d+) us', line)
    if not m:
        continue
    task = m.group(1)
    stages[(task, 'total')].append(int(m.group(3)))
    for name, us in re.findall(r'(\w+) (\d+)', m.group(2)):
        stages[(task, name)].append(int(us))
print(f'threads {os.environ.get("RC_THREADS", "auto")}, {N} frames, microseconds: median (min .. max)')
for (task, name), v in stages.items():
    print(f'  {task:16s} {name:8s} {statistics.median(v):6.0f}  ({min(v)} .. {max(v)})')

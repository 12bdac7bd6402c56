"""Which part of a frame step grows in the slow steps?  (round 5: ~3 of 20 timed steps of `bench.py` take 7-8 ms instead of 5.55.)
Runs the bench's frame step N times with the coder's TIMELINE marks on and, per step, the caching allocator's device-malloc counters,
the cyclic GC's collections, involuntary context switches and the per-window host times; prints the slow steps beside the median."""
import argparse, gc, os, resource, statistics, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=120)
ap.add_argument('--sleep-ms', type=float, default=0.0, help='idle time between steps')
ap.add_argument('--no-freeze', action='store_true')
args = ap.parse_args()
dev = torch.device('cuda:0')
import pcgcv2_amd
pcgcv2_amd.configure_host_threads(local_world=1, pin_device=None)
from pcgcv2_amd import synthetic, entropy_model
from pcgcv2_amd import coder as coder_mod
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.sparse import SparseTensor

p = synthetic.shell('shell10', device=dev)
c = torch.cat([torch.zeros((len(p), 1), dtype=torch.int32, device=dev), p], 1).contiguous()
x = SparseTensor(torch.ones((len(p), 1), dtype=torch.float32, device=dev), coordinates=c, tensor_stride=1, device=dev)
model = PCCModel().to(dev)
model.load_state_dict(synthetic.synthetic_state_dict())
tmp = tempfile.mkdtemp(prefix='pcgc_outl_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
coder = Coder(model, os.path.join(tmp, 'u'))
def cpu_stat():
    for path in ('/sys/fs/cgroup/cpu.stat', '/sys/fs/cgroup/cpu/cpu.stat'):
        try:
            return {k: int(v) for k, v in (l.split() for l in open(path).read().splitlines())}
        except OSError:
            pass
    return {}
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
print('cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
gcs = []
gc.callbacks.append(lambda phase, info: gcs.append((phase, info.get('generation'), time.perf_counter())) if phase == 'stop' else None)

def step(rec=None):
    x.cmap.drop_caches()
    coder_mod.TIMELINE = marks = []
    a = time.perf_counter()
    entropy_model.table_cache(clear=True)
    coder.encode(x, postfix='_s')
    b = time.perf_counter()
    entropy_model.table_cache(clear=True)
    out = coder.decode(postfix='_s')
    c_ = time.perf_counter()
    torch.cuda.synchronize()
    d = time.perf_counter()
    coder_mod.TIMELINE = None
    if rec is not None:
        m = dict(marks)
        rec.append({'total': (d - a) * 1e3, 'enc_win': (m['enc_gpu_done'] - a) * 1e3, 'enc_host_tail': (b - m['enc_gpu_done']) * 1e3,
                    'dec_to_first': (m['dec_gpu_first'] - b) * 1e3, 'dec_enqueue': (c_ - m['dec_gpu_first']) * 1e3, 'dec_drain': (d - c_) * 1e3})
    return out

for _ in range(8):
    step()
torch.cuda.synchronize()
if not args.no_freeze:
    gc.collect(); gc.freeze()
rows = []
for i in range(args.steps):
    st0 = torch.cuda.memory_stats(dev)
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    g0 = len(gcs)
    cs0 = cpu_stat()
    rec = []
    step(rec)
    st1 = torch.cuda.memory_stats(dev)
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    r = rec[0]
    r['dev_alloc'] = st1['num_device_alloc'] - st0['num_device_alloc']
    r['dev_free'] = st1['num_device_free'] - st0['num_device_free']
    r['retries'] = st1['num_alloc_retries'] - st0['num_alloc_retries']
    r['gc'] = [g[1] for g in gcs[g0:]]
    r['nivcsw'] = ru1.ru_nivcsw - ru0.ru_nivcsw
    r['minflt'] = ru1.ru_minflt - ru0.ru_minflt
    cs1 = cpu_stat()
    r['throttled'] = cs1.get('nr_throttled', 0) - cs0.get('nr_throttled', 0)
    r['throttled_us'] = cs1.get('throttled_usec', cs1.get('throttled_time', 0)) - cs0.get('throttled_usec', cs0.get('throttled_time', 0))
    r['cpu_ms'] = (ru1.ru_utime + ru1.ru_stime - ru0.ru_utime - ru0.ru_stime) * 1e3
    rows.append(r)
    if args.sleep_ms:
        time.sleep(args.sleep_ms / 1e3)
keys = ['total', 'enc_win', 'enc_host_tail', 'dec_to_first', 'dec_enqueue', 'dec_drain']
med = {k: statistics.median(r[k] for r in rows) for k in keys}
print('median      ' + '  '.join(f'{k} {med[k]:.2f}' for k in keys))
print(f'mean total {statistics.mean(r["total"] for r in rows):.3f} ms; reserved {torch.cuda.memory_reserved(dev) >> 20} MiB; '
      f'steps with device mallocs: {sum(1 for r in rows if r["dev_alloc"])}, with frees: {sum(1 for r in rows if r["dev_free"])}, with gc: {sum(1 for r in rows if r["gc"])}')
slow = [(i, r) for i, r in enumerate(rows) if r['total'] > 1.12 * med['total']]
print(f'{len(slow)} of {len(rows)} steps above 1.12 x median:')
for i, r in slow[:40]:
    print(f'  step {i:3d} ' + '  '.join(f'{k} {r[k]:.2f}' for k in keys) + f'  dev_alloc {r["dev_alloc"]} dev_free {r["dev_free"]} retries {r["retries"]} gc {r["gc"]} nivcsw {r["nivcsw"]} throttled {r["throttled"]} ({r["throttled_us"]} us) cpu {r["cpu_ms"]:.1f} ms')
print(f'process CPU time per step: median {statistics.median(r["cpu_ms"] for r in rows):.1f} ms; throttle events in the run: {sum(r["throttled"] for r in rows)}')
print('all totals: ' + ' '.join(f'{r["total"]:.2f}' for r in rows))

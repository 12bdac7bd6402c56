"""Multi-GPU sharding of the encode/decode path (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The reference is single-device (coder.py:5).  Frames — and octant blocks of one large cloud — are coded independently
(each unit is its own `Coder.encode/decode` with its own `postfix` file set, the mechanism test.py:38 uses for rates), so
the data path needs NO collective.  The only exchange is the final reduction of five scalars per rank
    [bits, N_in, N_out, sum d^2(A->B), sum d^2(B->A)]
for the aggregate bpp / D1 (one all-reduce of 40 bytes), plus an optional variable-length all-gather of decoded
coordinates when an exact global D1 over block borders is wanted.
"""
import math
import numpy as np
import torch
import torch.distributed as dist


FORCE_COLLECTIVES = False      # True: issue the collectives even in a one-rank group (tests: RCCL with world_size 1 on one GPU)


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _collectives_on():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def shard_units(n_units, rank=None, world_size=None):
    """Round-robin assignment of independent units (frames / blocks) to ranks -> list of unit indices of this rank."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return list(range(rank, n_units, world_size))


def split_octants(coords, levels=1, align=8):
    """Split a cloud into 8**levels spatial blocks for block-parallel coding (BASELINE config 5: 8 blocks, one per GPU).
    Each level halves the set along x, then y, then z at the MEDIAN coordinate rounded to a multiple of `align` (= the
    coarsest tensor stride, so no stride-8 cell is shared by two blocks): blocks are balanced in point count whatever
    the cloud's position inside its cube.  coords: int array/tensor [N,3] or [N,4] (batch first).
    -> list of index arrays (numpy, ascending), empty blocks dropped."""
    c = coords.detach().cpu().numpy() if isinstance(coords, torch.Tensor) else np.asarray(coords)
    xyz = c[:, -3:].astype(np.int64)
    parts = [np.arange(len(xyz))]
    for _ in range(levels):
        for d in range(3):
            nxt = []
            for idx in parts:
                if len(idx) == 0:
                    continue
                v = xyz[idx, d]
                cut = int(np.median(v))
                cut = ((cut + align // 2) // align) * align          # nearest lattice plane
                if cut <= v.min():
                    cut += align
                lo = v < cut
                nxt += [idx[lo], idx[~lo]]
            parts = nxt
    return [p for p in parts if len(p)]


class Stats:
    """Additive per-unit statistics; `reduce()` sums them over ranks (RCCL all-reduce on GPU, gloo on CPU)."""
    FIELDS = ('bits', 'n_in', 'n_out', 'sse_ab', 'sse_ba')

    def __init__(self, device='cpu'):
        self.v = torch.zeros(len(self.FIELDS), dtype=torch.float64, device=device)

    def add(self, bits=0, n_in=0, n_out=0, sse_ab=0.0, sse_ba=0.0):
        self.v += torch.tensor([bits, n_in, n_out, sse_ab, sse_ba], dtype=torch.float64, device=self.v.device)
        return self

    def reduce(self):
        if _collectives_on():
            dist.all_reduce(self.v, op=dist.ReduceOp.SUM)
        return self

    def summary(self, res):
        bits, n_in, n_out, ab, ba = [float(x) for x in self.v.tolist()]
        mse = max(ab / max(n_in, 1), ba / max(n_out, 1))
        peak = float(res - 1)
        return {'bits': bits, 'n_in': int(n_in), 'n_out': int(n_out), 'bpp': bits / max(n_in, 1),
                'd1_mse': mse, 'd1_psnr': float(10 * math.log10(3 * peak * peak / mse)) if mse > 0 else float('inf')}


def gather_varlen(rows, dst=0):
    """All ranks contribute an int32 [n_i, C] tensor; rank `dst` gets the concatenation (others get None).
    Sizes are exchanged first, then one padded all-gather (xGMI: a single large transfer beats many small ones)."""
    rank, w = world()
    if not _collectives_on():
        return rows
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    sizes = [torch.zeros_like(n) for _ in range(w)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = torch.zeros((m, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    pad[:rows.shape[0]] = rows
    bufs = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(bufs, pad)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0)


def _code_one(coder, name, x, rho, res, with_d1):
    from .pc_error import d1_psnr_device
    from .coder import stream_bits
    post = '_' + str(name)
    coder.encode(x, postfix=post)
    out = coder.decode(rho=rho, postfix=post)
    bits = int(stream_bits(coder.filename, post).sum())
    ab = ba = 0.0
    if with_d1:                                           # exact nearest neighbours on the GPU (pcgc_d1_nn)
        m = d1_psnr_device(x.C, out.C, res)
        ab, ba = m['sse1'], m['sse2']
    return out, dict(bits=bits, n_in=len(x), n_out=len(out), sse_ab=ab, sse_ba=ba)


def code_units(coder, units, rho=1.0, res=1024, with_d1=False, in_flight=1):
    """Encode+decode this rank's share of `units` = list of (name, SparseTensor) and return (Stats, {name: decoded}).
    Each unit uses postfix '_<name>' so its four files never collide.

    in_flight > 1 (serving / throughput mode): that many host threads, each with its own Coder and its own HIP stream, code
    different units concurrently.  Units are independent, so results are identical to the sequential order; what changes is
    that one unit's sequential host stages (range coder, octree coder, file I/O) and its small-level kernels overlap with
    the other units' GPU work — on `shell10`-class frames 4 in flight give ~1.8x the single-frame rate on one MI355X."""
    my = [units[i] for i in shard_units(len(units))]
    stats = Stats(device=units[0][1].device if units else 'cpu')
    outs = {}
    if in_flight <= 1 or len(my) <= 1:
        for name, x in my:
            outs[name], st = _code_one(coder, name, x, rho, res, with_d1)
            stats.add(**st)
        return stats, outs
    import queue
    import threading
    from .coder import Coder
    dev = my[0][1].device
    todo = queue.Queue()
    for u in my:
        todo.put(u)
    done, errors, lock = [], [], threading.Lock()
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(dev))          # the units' tensors were produced on the caller's stream (of their device)

    def worker():
        try:
            torch.cuda.set_device(dev)                    # current device and stream are per thread
            stream = torch.cuda.Stream(device=dev)
            stream.wait_event(ready)
            mine = Coder(coder.model, coder.filename)     # own staging buffers / side stream; files differ by postfix
            with torch.cuda.stream(stream):
                while True:
                    try:
                        name, x = todo.get_nowait()
                    except queue.Empty:
                        break
                    out, st = _code_one(mine, name, x, rho, res, with_d1)
                    with lock:
                        done.append((name, out, st))
                stream.synchronize()
        except BaseException as e:                        # surfaced to the caller below
            with lock:
                errors.append(e)

    threads = [threading.Thread(target=worker, name=f'pcgc-frame{i}') for i in range(min(in_flight, len(my)))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    order = {name: i for i, (name, _) in enumerate(my)}
    for name, out, st in sorted(done, key=lambda r: order[r[0]]):       # deterministic accumulation order
        outs[name] = out
        stats.add(**st)
    return stats, outs

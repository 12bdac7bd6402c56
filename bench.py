#!/usr/bin/env python3
"""bench.py — encode+decode throughput of the PCGCv2 hot path on MI355X.

Metric (BASELINE.json): encode+decode Mpoints/s at fixed rate (r3 stand-in: synthetic weights), vox10 frame.
`--config` picks the BASELINE.json configuration (the headline, and the default, is `frame`):

  frame   config 2: one vox10 frame per GPU per step — shell10 (786 632 points), the synthetic stand-in for
          longdress_vox10_1300.ply (real PLYs / checkpoints are external downloads).  A step = one `Coder.encode` + one
          `Coder.decode` including the four bitstream files: exactly what coder.py:155-162 brackets.
  batch4  config 3: the four vox10 frames shell10 / _b / _c / _d (8iVFB 4-sequence stand-ins), round-robin over the ranks;
          a step = every frame of the batch encoded + decoded once.
  sweep   config 4: shell11 (2.6 M points, dancer_vox11 stand-in) through 7 synthetic "rates" (latent gains), geometry maps
          shared by the rates as in pcgcv2_amd/test.py; a step = the whole 7-rate sweep (reference: 17.81 s,
          results/dancer_vox11_00000001.csv).
  blocks  config 5: shell12 (4.8 M points, House_without_roof vox12 stand-in) scaled by 0.375 and split into 8 octant
          blocks, round-robin over the ranks; a step = every block encoded + decoded once.

The input sparse tensors are resident in HBM when the timer starts; every coordinate level / kernel map / hash table derived
from them is dropped before each step (except inside `sweep`, whose point is the reuse across rates), so each step rebuilds
the whole geometry pyramid.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL); frames / blocks are independent units, so ranks
shard them with no data-path collective ("weak" scaling for `frame`: 1 frame per GPU per step); the only collective is the
final all-reduce of scalars after the timed region.

JSON line: the round contract plus `roofline` (dominant sparse-conv kernel: found by an untimed analysis step that brackets
every sparse-conv launch with HIP events, then bracketed alone inside the timed region) and `cpu_baseline` (the CPU oracle
timed on this host on the SAME workload, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
SWEEP_GAINS = (8.0, 16.0, 30.0, 50.0, 80.0, 120.0, 200.0)      # latent gains of the 7 synthetic "rates" (alphabets ~7..160 symbols)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='frame', choices=['frame', 'batch4', 'sweep', 'blocks'])
    ap.add_argument('--workload', default='', help='override the synthetic cloud of the chosen config (e.g. shell9 for a quick run)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', default='', help='cloud for the CPU oracle (default: the bench workload itself)')
    ap.add_argument('--no-events', action='store_true', help='do not bracket the dominant kernel with HIP events')
    ap.add_argument('--detail', default='', help='optional path for a per-kernel-shape JSON breakdown')
    ap.add_argument('--no-extra', action='store_true', help='skip the auxiliary measurements after the timed region (reference-format-only / one-by-one / serving): profiler runs')
    ap.add_argument('--no-batch', action='store_true', help='blocks config: code the blocks one by one instead of as ONE collated batch')
    ap.add_argument('--serving-frames', type=int, default=16, help='frames of the extra serving-throughput measurement (0 = skip; frame config only)')
    ap.add_argument('--serving-in-flight', type=int, default=4, help='frames in flight per GPU in that measurement')
    ap.add_argument('--warm-tables', action='store_true', help='keep the CDF-table caches across steps (the round-3 behaviour); default: every encode and '
                                                               'every decode evaluates its table, as the reference does')
    ap.add_argument('--two-cpu-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--input-order', default='raster', choices=['raster', 'shuffled'], help='row order of the input cloud (frame config)')
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with `python -m torch.distributed.run --nnodes=1 '
              f'--nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...` (one rank per GPU)', file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print('bench.py: no ROCm device visible; the codec has no CPU path', file=sys.stderr)
        sys.exit(3)
    local = local % max(1, torch.cuda.device_count())      # (test mode: several ranks may share one GPU)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    backend = os.environ.get('PCGC_DIST_BACKEND', 'nccl')      # 'nccl' = RCCL over xGMI; 'gloo' only to test the path on 1 GPU
    # PCGC_DIST_FORCE=1: create the process group and issue every collective even with ONE rank, so that the RCCL path (communicator
    # set-up, the two all-reduces, the padded all-gather of decoded coordinates) executes on the device where only one GPU exists
    dist_on = world > 1 or os.environ.get('PCGC_DIST_FORCE', '0') == '1'
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = dev if backend == 'nccl' else torch.device('cpu')

    import pcgcv2_amd
    # one node: the ranks share its CPUs (quota-aware pool sizes); with several ranks each is pinned to its GPU's NUMA-local CPUs
    host_threads = pcgcv2_amd.configure_host_threads(local_world=world, pin_device=local if world > 1 else None)
    from pcgcv2_amd import synthetic, ops, shard
    shard.FORCE_COLLECTIVES = dist_on and world == 1
    from pcgcv2_amd.pcc_model import PCCModel
    from pcgcv2_amd.coder import Coder, STREAMS
    from pcgcv2_amd import coder as coder_mod
    from pcgcv2_amd.data_utils import scale_sparse_tensor
    from pcgcv2_amd.sparse import SparseTensor
    from pcgcv2_amd import entropy_model

    def cloud(name, order='raster'):
        if name in synthetic.SHELLS and order == 'raster':
            p = synthetic.shell(name, device=dev)
        else:
            p = synthetic.cloud(name, order=order, seed=1).to(dev)
        c = torch.cat([torch.zeros((len(p), 1), dtype=torch.int32, device=dev), p], 1).contiguous()
        return SparseTensor(torch.ones((len(p), 1), dtype=torch.float32, device=dev), coordinates=c, tensor_stride=1, device=dev)

    # ---- the units of one step (this rank's share) ----
    cfg = args.config
    base = args.workload or {'frame': 'shell10', 'batch4': 'shell10', 'sweep': 'shell11', 'blocks': 'shell12'}[cfg]
    variants = [base] + [base + s for s in ('_b', '_c', '_d') if base + s in synthetic.SHELLS]
    sd = synthetic.synthetic_state_dict()
    model = PCCModel().to(dev)
    model.load_state_dict(sd)
    rate_sds = None
    if cfg == 'frame':                       # one frame per rank (distinct clouds per rank, like the 8iVFB 4-sequence config)
        units = [(variants[rank % len(variants)], cloud(variants[rank % len(variants)], args.input_order))]
        scaling = 'weak'
        desc = f'{base}: ' + ('perturbed-sphere vox10 frame' if base in synthetic.SHELLS else 'synthetic cloud') + f', 1 frame per GPU per step, rows in {args.input_order} order'
    elif cfg == 'batch4':
        mine = shard.shard_units(len(variants), rank, world)
        units = [(variants[i], cloud(variants[i])) for i in mine]
        scaling = 'strong'
        desc = f'{"+".join(variants)}: 4-frame vox10 batch, frames round-robin over {world} GPU(s)'
    elif cfg == 'sweep':
        units = [(base, cloud(base))]                  # every rank sweeps its own copy of the frame (weak scaling)
        # one checkpoint per rate, staged on the device before the clock starts (test.py loads its .pth files outside the timers too):
        # inside a step a rate change is 227 device-to-device parameter copies, not 227 host-to-device ones
        rate_sds = [{k: v.to(dev) for k, v in synthetic.synthetic_state_dict(gain=g).items()} for g in SWEEP_GAINS]
        scaling = 'weak'
        desc = f'{base}: vox11 frame through {len(SWEEP_GAINS)} synthetic rates (latent gains {SWEEP_GAINS}), geometry maps shared by the rates'
    else:
        whole = cloud(base)
        x_in = scale_sparse_tensor(whole, 0.375)                               # data_utils.py:112-118
        blocks = shard.split_octants(x_in.C, levels=1)
        mine = shard.shard_units(len(blocks), rank, world)
        units = []
        for i in mine:
            c = x_in.C[torch.as_tensor(blocks[i], device=dev)]
            units.append((f'b{i}', SparseTensor(torch.ones((len(c), 1), device=dev), coordinates=c, tensor_stride=1, device=dev, assume_unique=True)))
        scaling = 'strong'
        desc = (f'{base}: vox12 cloud ({len(whole)} points) scaled by 0.375 -> {len(x_in)} points, {len(blocks)} octant blocks '
                f'round-robin over {world} GPU(s), this rank\'s blocks collated into ONE batch per step (Coder.encode_batch / decode_batch: '
                'files and decoded voxels identical to coding them one by one)' if not args.no_batch else
                f'{base}: vox12 cloud ({len(whole)} points) scaled by 0.375 -> {len(x_in)} points, {len(blocks)} octant blocks round-robin over {world} GPU(s), coded one by one')
        whole_C = whole.C
        del whole
    n_points = sum(len(u) for _, u in units)
    # blocks: the rank's blocks as one collated batch (item index in column 0), resident in HBM like the single units
    batch = None
    if cfg in ('blocks', 'batch4') and len(units) > 1 and not args.no_batch:
        cb = torch.cat([torch.cat([torch.full((len(u), 1), i, dtype=torch.int32, device=dev), u.C[:, 1:]], 1) for i, (_, u) in enumerate(units)], 0).contiguous()
        batch = (SparseTensor(torch.ones((len(cb), 1), device=dev), coordinates=cb, tensor_stride=1, device=dev, assume_unique=True),
                 [f'_{name}' for name, _ in units])
    tmp = tempfile.mkdtemp(prefix=f'pcgc_bench_r{rank}_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    coder = Coder(model, os.path.join(tmp, 'u'))

    # CDF tables: a table is a pure function of (entropy parameters, symbol range) and this benchmark re-codes the same frame with the
    # same weights, so with the caches on (entropy_model.TABLE_CACHE, pcgc_table_cache) no timed step would ever evaluate one.  The
    # reference evaluates the table in EVERY compress and EVERY decompress (entropy_model.py:165-171,185-190), and a real decoder does not
    # share a process with the encoder: the headline therefore drops the caches before every encode and before every decode (inside the
    # timed region; the drop itself is a mutex and a dict clear).  `config.table_cache_warm` reports the cached figure beside it.
    cold = {'on': not args.warm_tables}

    def cold_tables():
        if cold['on']:
            entropy_model.table_cache(clear=True)

    def step(timers=None, one_by_one=False, these=None):
        """one pass over this rank's units -> (decoded tensors); timers = [enc_s, dec_s] accumulators"""
        outs = []
        if batch is not None and not one_by_one and these is None:
            xb, posts = batch
            xb.cmap.drop_caches()
            a = time.perf_counter()
            cold_tables()
            coder.encode_batch(xb, posts)
            if timers is not None:
                torch.cuda.synchronize()
            b = time.perf_counter()
            cold_tables()
            outs = coder.decode_batch(posts)
            if timers is not None:
                torch.cuda.synchronize()
                c = time.perf_counter()
                timers[0] += b - a
                timers[1] += c - b
            return outs
        for name, x in (units if these is None else these):
            x.cmap.drop_caches()
            rates = rate_sds if rate_sds is not None else [None]
            for ri, rsd in enumerate(rates):
                if rsd is not None:
                    model.load_state_dict(rsd)
                post = f'_{name}' + (f'_r{ri + 1}' if rsd is not None else '')
                a = time.perf_counter()
                cold_tables()
                coder.encode(x, postfix=post)
                if timers is not None:
                    torch.cuda.synchronize()
                b = time.perf_counter()
                cold_tables()
                outs.append(coder.decode(postfix=post))
                if timers is not None:
                    torch.cuda.synchronize()
                    c = time.perf_counter()
                    timers[0] += b - a
                    timers[1] += c - b
        return outs

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    outs = None
    outs = step()                                    # (first touch: allocator, tables of the dispatcher, LDS limits of the kernels)
    torch.cuda.synchronize()
    # Untimed analysis passes (not counted as warmup; they run BEFORE the W warm-up steps, so that the warm-up is what immediately precedes
    # the timed region — the passes below read maps back to the host and leave the GPU idle for milliseconds at a time): one step with EVERY sparse-conv launch bracketed by HIP events — that
    # finds the dominant (kernel, level) and gives the all-launch aggregate — and one that counts the kernel-map pairs per level
    # for the byte / flop formulas.  The timed region then brackets only the dominant kernel's launches, so the event overhead
    # (~0.6 ms per step with every launch bracketed) stays out of `value`.
    ANALYSIS_STEPS = 3
    dominant, dominant_keys, warm_all, warm_detail, pairs, families = None, (), None, None, None, None
    if not args.no_events and units:
        ops.PROFILE.reset(enabled=True)
        for _ in range(ANALYSIS_STEPS):              # three bracketed steps: the dominant kernel NAME is picked by median launch time x launches
            outs = step()
        ops.PROFILE.enabled = False
        ops.PROFILE.counting = True
        step()
        ops.PROFILE.counting = False
        torch.cuda.synchronize()
        dominant, dominant_keys = ops.PROFILE.dominant(ANALYSIS_STEPS, HBM_PEAK_GBS)
        warm_all = ops.PROFILE.aggregate(HBM_PEAK_GBS, ANALYSIS_STEPS)
        families = ops.PROFILE.families(ANALYSIS_STEPS, HBM_PEAK_GBS)
        warm_detail = [{k: v for k, v in d.items() if k != 'key'} for d in ops.PROFILE.detail()]
        pairs = dict(ops.PROFILE.pairs)
    elif outs is None:
        outs = step()
    # Python's cyclic GC stays enabled, but the ~1e6 long-lived objects created by importing torch & friends are moved
    # to the permanent generation so that full collections do not re-traverse them (tens of ms each) mid-measurement.
    import gc
    gc.collect()
    gc.freeze()
    n_out = sum(len(o) for o in outs)
    n_coded = n_points * (len(rate_sds) if rate_sds is not None else 1)        # points through encode+decode per step
    bits = index_bits = 0
    for f in os.listdir(tmp):
        if f.endswith(STREAMS):
            bits += os.path.getsize(os.path.join(tmp, f)) * 8
        elif f.endswith('_F.idx'):                              # decoding index of `_F.bin` (not one of the reference's four files)
            index_bits += os.path.getsize(os.path.join(tmp, f)) * 8

    # ---- timed region: exactly K steps, with the dominant kernel bracketed by HIP events on its own stream ----
    ops.PROFILE.reset(enabled=dominant is not None, only=dominant_keys)
    if dominant is not None:
        ops.PROFILE.pairs = pairs
    # the W warm-up steps, under exactly the timed region's instrumentation (the pair-counting pass above leaves allocator and caches in a
    # state no timed step ever sees: without a step in between the first timed step cost +1 ms)
    for _ in range(args.warmup):
        step()
    if dominant is not None:
        ops.PROFILE.reset(enabled=True, only=dominant_keys)
        ops.PROFILE.pairs = pairs
    timers = [0.0, 0.0]
    step_ms = []
    barrier()
    t0 = time.perf_counter()
    # The dominant kernel's launches are bracketed in every EVENT_EVERY-th timed step only: a bracket is two hipEventRecords with timing, i.e. two
    # barrier packets around the launch, and the name-pooled dominant kernel has six launches per step — bracketing all of them in every step cost
    # the step 0.3 ms (5 %: the headline ran that much slower than the unbracketed auxiliary measurements of the same process).
    EVENT_EVERY = 4
    for i_step in range(args.steps):
        ops.PROFILE.enabled = dominant is not None and i_step % EVENT_EVERY == 0
        a = time.perf_counter()
        step(timers)
        step_ms.append(round((time.perf_counter() - a) * 1e3, 2))
    barrier()
    elapsed = time.perf_counter() - t0
    ops.PROFILE.enabled = False
    torch.cuda.synchronize()
    bracketed_steps = len(range(0, args.steps, EVENT_EVERY))
    roof = ops.PROFILE.summary(HBM_PEAK_GBS, bracketed_steps, name=dominant) if dominant is not None else None
    if roof is not None:
        roof['sampling'] = (f'launches of the dominant kernel bracketed with HIP events (on the stream they are launched on) in every {EVENT_EVERY}th step of the timed '
                            f'region: {bracketed_steps} of {args.steps} steps, {roof["launches_timed"]} launches')
    if roof is not None and warm_all is not None:
        # the dominant kernel's own figures above are LIVE (its launches bracketed inside the timed region); the all-launch aggregate and the
        # per-kernel table come from the untimed analysis steps, where EVERY sparse-conv launch is bracketed (that costs ~0.6 ms per step)
        roof['all_launches'] = dict(warm_all, measured=f'{ANALYSIS_STEPS} untimed analysis steps before the timed region, every sparse-conv launch bracketed '
                                                       '(k3 / k1 convs, fused InceptionResNet passes, k2 s2 down convs, generative up convs)')
        roof['families'] = families

    # the same K steps on the reference's four files alone (no `_F.idx` sidecar: the feature stream is decoded serially, exactly as a
    # reference-made stream would be) — reported beside `value`, so rate and speed of BOTH configurations are on the line
    plain = None
    if cfg == 'frame' and not args.no_extra:
        keep_segments, coder_mod.INDEX_SEGMENTS = coder_mod.INDEX_SEGMENTS, 0
        try:
            step()
            barrier()
            t_p = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            dt_p = time.perf_counter() - t_p
        finally:
            coder_mod.INDEX_SEGMENTS = keep_segments
        plain = {'value': round(n_coded * args.steps / dt_p / 1e6, 4), 'unit': 'Mpoints/s', 'ms_per_step': round(dt_p / args.steps * 1e3, 3),
                 'note': 'this rank, the reference format only (INDEX_SEGMENTS = 0: no sidecar written or read, `_F.bin` decoded by one thread)'}
        step()                                           # leave the files of the default configuration behind

    def timed_steps(**kw):
        for _ in range(max(1, args.warmup)):                     # (the auxiliary figures get the headline's warm-up: one step left them 5-10 % noise)
            step(**kw)
        barrier()
        t_p = time.perf_counter()
        for _ in range(args.steps):
            step(**kw)
        barrier()
        return time.perf_counter() - t_p

    # the same K steps with the table caches left warm (what round 3 timed): reported beside the cold headline
    warm_tables = None
    if not args.no_extra and cold['on']:
        cold['on'] = False
        try:
            dt_p = timed_steps()
        finally:
            cold['on'] = True
        warm_tables = {'value': round(n_coded * args.steps / dt_p / 1e6, 4), 'unit': 'Mpoints/s', 'ms_per_step': round(dt_p / args.steps * 1e3, 3),
                       'note': 'this rank, CDF-table caches kept across steps: after warm-up no encode or decode evaluates a table (a property of re-coding '
                               'one frame in one process; `value` drops the caches before every encode and every decode)'}

    # the headline cloud with its rows in the OTHER order: the canonical row order of every encoder level follows the input order
    # (a raster-ordered synthetic shell is the friendliest case for the gather kernels; a scanner's PLY has no particular order)
    order_sens = None
    if cfg == 'frame' and not args.no_extra:
        other = 'shuffled' if args.input_order == 'raster' else 'raster'
        name0 = units[0][0]
        alt = [(name0 + '_alt', cloud(name0, other))]
        dt_p = timed_steps(these=alt)
        order_sens = {args.input_order: None, other: round(len(alt[0][1]) * args.steps / dt_p / 1e6, 4), 'unit': 'Mpoints/s',
                      'note': f'this rank: the same cloud with its rows in {other} order (numpy default_rng(1) permutation), same K steps, cold tables; '
                              'the bitstream is identical (the latent is sorted before coding), the decoder works in the sorted latent\'s order either way — '
                              'only the encoder sees the input order'}
        del alt
        step()

    # the same K steps on geometry that is NOT a smooth closed surface: holes / drop-outs / salt (noisy10), intersecting
    # shells + one-voxel sheets + a rod + a solid block (multi10), a filled body with 27 / 27 neighbourhoods and whole regions of tied
    # logits (solid_ball); cold tables, the headline's warm-up
    geom_sens = None
    if cfg == 'frame' and world == 1 and not args.no_extra and not args.two_cpu_child and not args.workload:
        geom_sens = {}
        for nm in ('noisy10', 'multi10', 'solid_ball'):
            alt = [(nm, cloud(nm))]
            dt_p = timed_steps(these=alt)
            geom_sens[nm] = {'points': int(len(alt[0][1])), 'Mpoints_s': round(len(alt[0][1]) * args.steps / dt_p / 1e6, 4),
                             'ms_per_step': round(dt_p / args.steps * 1e3, 3)}
            del alt
        geom_sens['note'] = ('this rank, the same K steps (cold tables) on synthetic.CLOUDS; the headline cloud (shell10) is `value`; every figure codes '
                             'ONE frame at a time, so smaller clouds pay the same ~1.3 ms host-serial window over fewer points')
        step()

    # where a step's wall-clock goes: five untimed steps with two host marks inside the coder — after the encoder's
    # symbols have reached the host (nothing is queued on the GPU any more) and before the decoder's first device work is enqueued.  Between
    # the two the GPU idles behind the sequential host stages (range encoder, files, file read, coordinate decode, table); outside them the
    # host launches and the GPU executes concurrently.
    def step_windows(these=None):
        spans = []
        for _ in range(5):
            coder_mod.TIMELINE = marks = []
            torch.cuda.synchronize()
            t_a = time.perf_counter()
            step(these=these)
            torch.cuda.synchronize()
            t_b = time.perf_counter()
            coder_mod.TIMELINE = None
            m = dict(marks)
            if 'enc_gpu_done' in m and 'dec_gpu_first' in m:
                spans.append(((m['enc_gpu_done'] - t_a) * 1e3, (m['dec_gpu_first'] - m['enc_gpu_done']) * 1e3, (t_b - m['dec_gpu_first']) * 1e3))
        if not spans:
            return None
        med = lambda i: round(sorted(sp[i] for sp in spans)[len(spans) // 2], 3)
        return {'enc_gpu_window_ms': med(0), 'host_serial_ms': med(1), 'dec_gpu_window_ms': med(2)}

    windows = None
    if cfg == 'frame' and not args.no_extra and batch is None and rate_sds is None:
        windows = step_windows()
        if windows:
            windows.update(gpu_conv_ms=None if warm_all is None else warm_all['ms_per_step'],
                           note='medians of 5 untimed steps (host clock, no extra device synchronisation inside the step).  enc_gpu_window: step start -> '
                                "the latent's symbols are on the host (pyramid, maps, encoder convolutions, sort, quantisation; the host enqueues ahead of the "
                                'GPU).  host_serial: -> the decoder\'s first device work is enqueued: range encoder, table, files, file read, coordinate-stream '
                                'decode — the GPU has NOTHING queued in this window.  dec_gpu_window: -> decode complete (level upload, feature-stream decode '
                                'finishing beside it, decoder convolutions, top-k / pruning).  gpu_conv_ms = sum of ALL sparse-conv launches of one step (HIP '
                                'events, analysis pass); the GPU-busy total incl. the non-conv kernels is in profiles/r06_kernel_trace.txt')
        if geom_sens:
            for nm in ('noisy10', 'multi10', 'solid_ball'):
                if nm in geom_sens:
                    geom_sens[nm]['step_windows'] = step_windows(these=[(nm, cloud(nm))])
            step()

    # one rank on a TWO-CPU budget (eight ranks of a node that share a 16-CPU quota get exactly that): the same command in a child process
    # whose affinity mask holds two CPUs — configure_host_threads then budgets one range-decoder thread (the lane-parallel decoder), no pools
    two_cpu = None
    if cfg == 'frame' and world == 1 and not args.no_extra and not args.two_cpu_child and hasattr(os, 'sched_setaffinity'):
        import subprocess
        cpus = sorted(os.sched_getaffinity(0))[:2]
        if len(cpus) == 2:
            torch.cuda.synchronize()
            cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(args.steps), '--warmup', str(args.warmup), '--no-extra', '--no-cpu-baseline',
                   '--no-events', '--two-cpu-child', '--input-order', args.input_order] + (['--warm-tables'] if args.warm_tables else []) + \
                  (['--workload', args.workload] if args.workload else [])
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, preexec_fn=lambda: os.sched_setaffinity(0, set(cpus)))
                child = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
                if child:
                    c = child[-1]
                    two_cpu = {'value': c['value'], 'unit': 'Mpoints/s', 'ms_per_step': c['ms_per_step'], 'enc_ms': c['config']['enc_ms'], 'dec_ms': c['config']['dec_ms'],
                               'host_threads': c['config']['host_threads'],
                               'note': f'the same K steps in a child process restricted to CPUs {cpus} (sched_setaffinity): one launcher thread + one helper; '
                                       'the feature stream is decoded by the lane-parallel decoder (eight segments per zmm register, no pool threads)'}
                else:
                    two_cpu = {'error': (r.stderr or r.stdout)[-300:]}
            except (subprocess.TimeoutExpired, OSError) as e:
                two_cpu = {'error': str(e)[:200]}

    one_by_one = None
    if batch is not None and not args.no_extra:
        step(one_by_one=True)
        barrier()
        t_p = time.perf_counter()
        for _ in range(args.steps):
            step(one_by_one=True)
        barrier()
        dt_p = time.perf_counter() - t_p
        one_by_one = {'value': round(n_coded * args.steps / dt_p / 1e6, 4), 'unit': 'Mpoints/s', 'ms_per_step': round(dt_p / args.steps * 1e3, 3),
                      'note': "this rank's blocks coded one after the other (one Coder.encode + decode per block): the latency-bound form"}
        step()

    serving = None
    if cfg == 'frame' and world == 1 and args.serving_frames > 0 and not args.no_extra:
        # serving mode (reported beside the headline, never as `value`): independent frames collated into batches of F and coded by ONE
        # encoder / decoder pass per batch (Coder.encode_batch / decode_batch) — one frame's sequential host stages (range coder, octree
        # coder, files) run beside the other frames' on a thread pool inside the call, the small-level kernels work on F-times larger
        # levels.  Bitstreams and decoded clouds are byte-identical to frame-by-frame coding (tests/test_gpu_parity.py).  (Round 2 ran F
        # host threads with a Coder and a HIP stream each: 34-165 Mpoints/s depending on how the box scheduled them; this form has no
        # thread race in it.)
        model.load_state_dict(sd)
        F = max(1, args.serving_in_flight)
        n_batches = max(1, args.serving_frames // F)
        batches = []
        for bi in range(n_batches):
            cs = [cloud(variants[(bi * F + i) % len(variants)]).C for i in range(F)]
            cbat = torch.cat([torch.cat([torch.full((len(c), 1), i, dtype=torch.int32, device=dev), c[:, 1:]], 1) for i, c in enumerate(cs)], 0).contiguous()
            batches.append((SparseTensor(torch.ones((len(cbat), 1), device=dev), coordinates=cbat, tensor_stride=1, device=dev, assume_unique=True),
                            [f'_s{bi}_{i}' for i in range(F)]))
        n_s = sum(len(xb) for xb, _ in batches)

        def serve():
            for xb, posts in batches:
                xb.cmap.drop_caches()                # no geometry survives between batches either
                coder.encode_batch(xb, posts)
                coder.decode_batch(posts)
        serve()
        torch.cuda.synchronize()
        dts = []
        for _ in range(3):                           # best of three passes (all reported): the figure is auxiliary
            t_s = time.perf_counter()
            serve()
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t_s)
        dt_s = min(dts)
        serving = {'frames_per_batch': F, 'frames': n_batches * F, 'value': round(n_s / dt_s / 1e6, 3), 'unit': 'Mpoints/s',
                   'passes_Mpoints_s': [round(n_s / d / 1e6, 1) for d in dts],
                   'note': 'independent vox10 frames collated F at a time and coded by one encoder / decoder pass per batch (Coder.encode_batch / '
                           'decode_batch), best of 3 passes; the headline `value` is the single-frame-at-a-time rate'}
        del batches
    elif cfg in ('batch4', 'blocks') and args.serving_in_flight > 1 and len(units) > 1 and batch is None:
        # the same units of this rank, several in flight (blocks / frames are independent; only the single-unit latency needs them one by one)
        for _, u in units:
            u.cmap.drop_caches()
        shard.code_units(coder, units, in_flight=args.serving_in_flight)
        torch.cuda.synchronize()
        dt_s = float('inf')
        for _ in range(2):
            for _, u in units:
                u.cmap.drop_caches()
            t_s = time.perf_counter()
            shard.code_units(coder, units, in_flight=args.serving_in_flight)
            torch.cuda.synchronize()
            dt_s = min(dt_s, time.perf_counter() - t_s)
        serving = {'units_in_flight': args.serving_in_flight, 'units': len(units), 'value': round(n_points / dt_s / 1e6, 3), 'unit': 'Mpoints/s',
                   'ms_per_step': round(dt_s * 1e3, 3),
                   'note': "this rank's units of one step coded concurrently (own thread + HIP stream each), best of 2 passes; `value` codes them one by one"}

    # the coordinate-coder stage on its own (report with and without it).  Inside a step it runs on a helper
    # thread concurrently with the GPU, so it adds nothing to ms_per_step unless it outlasts the work it hides behind.
    coord_ms = None
    if units:
        from pcgcv2_amd import gpcc
        first = [f for f in sorted(os.listdir(tmp)) if f.endswith('_C.bin')][0]
        c8 = gpcc.native_decode(os.path.join(tmp, first))
        t_c = time.perf_counter(); gpcc.native_encode(c8, os.path.join(tmp, 'probe_C.bin')); t_c1 = time.perf_counter()
        gpcc.native_decode(os.path.join(tmp, 'probe_C.bin')); t_c2 = time.perf_counter()
        coord_ms = {'encode': round((t_c1 - t_c) * 1e3, 3), 'decode': round((t_c2 - t_c1) * 1e3, 3), 'points': int(len(c8)),
                    'note': 'host octree coder alone (one unit); overlapped with GPU work inside a step'}

    # `_C.bin` on geometry that is NOT the shape the octree coder's contexts were trained on (they are trained, once per process, on an
    # integer-defined sphere shell; every bench cloud is a sphere shell too: its 1.7 bit per stride-8 point is a best case)
    coord_rate = None
    if cfg == 'frame' and world == 1 and not args.no_extra and not args.two_cpu_child:
        from pcgcv2_amd import gpcc
        from pcgcv2_amd.sparse import CoordMap
        coord_rate = {}
        for nm in ('shell10', 'noisy10', 'multi10', 'solid_ball'):
            p3 = (synthetic.shell(nm, device=dev) if nm in synthetic.SHELLS else synthetic.cloud(nm).to(dev))
            c4 = torch.cat([torch.zeros((len(p3), 1), dtype=torch.int32, device=dev), p3], 1).contiguous()
            l8 = CoordMap(c4, 1, unique=True).build_pyramid(3)
            c8 = (l8.C[:, 1:] // 8).cpu().numpy()
            path = os.path.join(tmp, f'rate_{nm}_C.bin')
            gpcc.native_encode(c8, path)
            back = gpcc.native_decode(path)
            assert len(back) == len(c8)
            coord_rate[nm] = {'stride8_points': int(len(c8)), 'bits_per_stride8_point': round(os.path.getsize(path) * 8 / max(len(c8), 1), 3),
                              'bpp_of_the_input_cloud': round(os.path.getsize(path) * 8 / max(len(p3), 1), 5)}
            os.remove(path)
        coord_rate['note'] = ('native octree stream (tmc3 absent; container version 5: contexts start from a prior trained on a sphere, an ellipsoid, a tilted plane and a '
                              'ragged noisy shell — none of these four clouds — and adapt fast on their first visits) on the stride-8 level of four clouds; the 8 independent '
                              'groups that keep the decode at 0.2 ms cost 0.27 bit per point against the one-stream form (profiles/r05_coord_codec.md)')

    enc_t, dec_t = timers
    if dist_on:
        t = torch.tensor([elapsed, enc_t, dec_t], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, enc_t, dec_t = [float(v) for v in t.tolist()]
        tot = torch.tensor([float(n_coded), float(n_out), float(bits), float(n_points), float(index_bits)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_coded, total_out, total_bits, total_points, total_index_bits = [float(v) for v in tot.tolist()]
    else:
        total_coded, total_out, total_bits, total_points, total_index_bits = float(n_coded), float(n_out), float(bits), float(n_points), float(index_bits)

    # quality of this rank's first unit (outside the timed region, as coder.py:180-182): D1 on the GPU
    d1 = None
    if units and cfg != 'blocks':
        from pcgcv2_amd.pc_error import d1_psnr_device
        d1 = d1_psnr_device(units[0][1].C, outs[0].C, {'shell11': 2048, 'shell12': 4096}.get(base, 1024))
    elif cfg == 'blocks':
        # exact GLOBAL D1 of the blocked cloud (SURVEY 8e caveat b: a per-block D1 misses nearest neighbours across block borders):
        # every rank's decoded blocks travel to rank 0 in one padded all-gather (shard.gather_varlen: RCCL over xGMI), are scaled back
        # (coder.py:166) and compared with the original cloud there
        from pcgcv2_amd.pc_error import d1_psnr_device
        mine_dec = torch.cat([o.C for o in outs], 0) if outs else torch.zeros((0, 4), dtype=torch.int32, device=dev)
        all_dec = shard.gather_varlen(mine_dec.contiguous(), dst=0)
        if rank == 0:
            dec = SparseTensor(torch.ones((len(all_dec), 1), device=dev), coordinates=all_dec, tensor_stride=1, device=dev)
            back = scale_sparse_tensor(dec, 1.0 / 0.375)
            d1 = d1_psnr_device(whole_C, back.C, 4096)
    if rank == 0:
        value = total_coded * args.steps / elapsed / 1e6
        nbar = None
        if roof is not None:
            big = max(warm_detail, key=lambda d: d['n_out'] if 'pairs' in d and d['name'].startswith(('k_child', 'k3', 'k_conv_packed', 'k_rows_irn', 'k_rows_conv')) else -1)
            nbar = round(big['pairs'] / big['n_out'], 2)
        line = {
            'metric': 'encode+decode Mpoints/sec at fixed bpp (r3 ckpt), vox10 frame',
            'value': round(value, 4), 'unit': 'Mpoints/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': scaling,
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{cfg}: {desc}; synthetic r3 stand-in weights (seed 1234, gain 50), encode+decode incl. bitstream files',
                       'baseline_config': {'frame': 2, 'batch4': 3, 'sweep': 4, 'blocks': 5}[cfg],
                       'points_in_per_step_all_gpus': int(total_points), 'points_coded_per_step_all_gpus': int(total_coded),
                       'enc_ms': round(enc_t / args.steps * 1e3, 3), 'dec_ms': round(dec_t / args.steps * 1e3, 3),
                       'enc_plus_dec_s_per_step': round((enc_t + dec_t) / args.steps, 4),
                       'bpp': round((total_bits + total_index_bits) / max(total_coded, 1), 5), 'points_out': int(total_out),
                       'bpp_note': '`bpp` counts everything the timed configuration writes: the reference\'s four files AND the `_F.idx` sidecar '
                                   '(decoding index + table guard); `bpp_reference_files_only` is the four files, the rate `reference_format_only` codes at',
                       'bpp_reference_files_only': round(total_bits / max(total_coded, 1), 5),
                       'reference_format_only': plain, 'units_one_by_one': one_by_one, 'host_threads': host_threads,
                       'table_mode': model.entropy_bottleneck.table_mode + (': the reference\'s arithmetic (ATen CPU operators issued from C++, csrc/reftable.cpp), NOT the fused HIP '
                                                                            'kernel pcgc_cdf_table (table_mode="device": <= 1 count off the reference, not interoperable); tables are '
                                                                            'host-kind dependent exactly as the reference\'s are' if model.entropy_bottleneck.table_mode == 'reference' else ''),
                       'table_cache': 'warm (caches kept across steps)' if args.warm_tables else 'cold: dropped before every encode and every decode inside the timed region '
                                      '(the reference evaluates a table per compress / decompress call)',
                       'table_cache_warm': warm_tables, 'input_order': args.input_order, 'input_order_sensitivity': order_sens, 'two_cpu_rank': two_cpu,
                       'geometry_sensitivity': geom_sens, 'step_windows': windows,
                       'entropy_decode': '`_F.bin` (bit-identical to the reference-format stream, decodable without it) comes with a sidecar '
                                         f'`_F.idx` of decoder states at {coder_mod.INDEX_SEGMENTS} row boundaries: its segments are decoded two per thread (two dependency chains per loop) on up to 8 threads; `_C.bin` '
                                         '(native octree, tmc3 absent) is coded as up to 8 independent groups of subtrees',
                       'path_switches': ops.PATH.switches(),
                       'coord_codec': 'native-octree (tmc3 absent)', 'coord_coder_ms': coord_ms, 'coord_codec_rate': coord_rate, 'serving_throughput': serving,
                       'step_ms_rank0': step_ms,
                       'd1_psnr_rank0_db': None if d1 is None else round(d1['mseF,PSNR (p2point)'], 4),
                       'd1_scope': 'whole blocked cloud, decoded blocks gathered to rank 0 (shard.gather_varlen)' if cfg == 'blocks' else "this rank's first unit",
                       'rccl_loaded': any('librccl' in l for l in open('/proc/self/maps')),
                       'collectives': ('RCCL' if backend == 'nccl' else backend) + (' (forced with one rank)' if dist_on and world == 1 else '') if dist_on else 'none (single process)',
                       'caveats': 'synthetic random weights: the decoder keeps the wrong voxels, so D1 only shows that the metric path runs, and bpp '
                                  '(~1.0) is ~11x the 0.093 of the real r3 checkpoint; the mean kernel-map occupancy of the dominant level is '
                                  f'{nbar} neighbours per row here against 18.7 probed for true geometry (SURVEY 8d) — a real checkpoint would see '
                                  '~20 % more pairs on the stride-1 level'},
            'roofline': roof,
        }
        if order_sens is not None:
            order_sens[args.input_order] = round(value, 4) if world == 1 else None
        if roof is not None and cfg == 'frame' and not args.workload:      # (the PMC passes were collected on the frame workload)
            attach_pmc_traffic(roof)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(cfg, args.cpu_sample or base, sd)
        if args.detail and warm_detail is not None:
            with open(args.detail, 'w') as f:
                json.dump(warm_detail, f, indent=1)
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.destroy_process_group()


def attach_pmc_traffic(roof):
    """`traffic`: HBM bytes per launch of the dominant kernel from rocprofv3 PMC counters (FETCH_SIZE and WRITE_SIZE in
    separate passes; FETCH_SIZE doubled for 16-byte-per-lane reads as MI355X_MICROARCH.md prescribes for gfx950).  The
    counters cannot be read from inside the process: the figure is REPLAYED from the committed collection of the same command
    (tools/pmc_traffic.sh -> profiles/pmc_traffic.json), not measured in this run; null if no entry matches."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if not os.path.exists(path):
        return
    try:
        table = json.load(open(path))
    except ValueError:
        return
    # the collection is only replayed while the kernel sources are the ones it was collected on (tools/pmc_traffic.py stamps their hash)
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'pcgcv2_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):            # device sources (the host codec, ply and table .cpp files launch nothing)
            h.update(f.encode()); h.update(open(os.path.join(d, f), 'rb').read())
    if table.get('kernel_sources_sha16') != h.hexdigest()[:16]:
        roof['traffic_source'] = ('profiles/pmc_traffic.json was collected on other kernel sources (stamp %s, now %s): not replayed; re-run tools/pmc_traffic.sh'
                                  % (table.get('kernel_sources_sha16'), h.hexdigest()[:16]))
        return
    short = roof['kernel'].split(' (')[0].rstrip('>')                     # e.g. "k_child_irn_a<16" or "k_rows_irn_a64"
    cands = []
    for e in table.get('kernels', []):
        name = e['kernel'].replace('(anonymous namespace)::', '').replace('void ', '')
        if name == short or name.startswith(short + ',') or name.startswith(short + '>') or name.startswith(short + '<'):
            cands.append(e)
    if not cands:
        return
    # the roofline object pools the kernel's levels, so does the traffic: mean over the sampled launches of every grid the name ran on
    n = sum(e['launches_sampled'] for e in cands)
    per_launch = sum(e['hbm_bytes_per_launch'] * e['launches_sampled'] for e in cands) / n
    roof['traffic'] = round(per_launch)
    roof['traffic_source'] = 'replayed from profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command); not measured in this run'
    roof['traffic_rate'] = {'GBps': round(per_launch / (roof['avg_launch_us'] * 1e-6) / 1e9, 1),
                            'frac_of_hbm_peak': round(per_launch / (roof['avg_launch_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                            'over_compulsory': round(per_launch / max(roof['algorithmic']['compulsory_bytes_per_launch'], 1), 3)}
    roof['traffic_detail'] = [{k: e[k] for k in ('kernel', 'grid_size', 'launches_sampled', 'fetch_bytes_raw', 'fetch_bytes_corrected', 'write_bytes') if k in e} for e in cands]


def cpu_baseline(cfg, sample, sd):
    """The CPU oracle (oracle/: C restatement, OpenMP over output rows) on the bench workload itself — one encode+decode of the
    same cloud with all the cores the container may use — and a one-thread figure on a smaller cloud (SURVEY §8d asks for both).
    kind = "port": MinkowskiEngine's CPU backend cannot be installed here (no network, un-vendored).  For the multi-unit
    configs the sample is one unit (one frame / the unscaled cloud's first block-sized piece is not split: the frame itself)."""
    import pcgcv2_amd
    from pcgcv2_amd import synthetic
    cores = pcgcv2_amd.effective_cpus()                     # cgroup quota, not os.cpu_count()
    os.environ['OMP_NUM_THREADS'] = str(cores)
    from oracle import pcgc_oracle as orc
    sd_np = synthetic.state_dict_to_numpy(sd)
    if cfg in ('sweep', 'blocks'):
        sample = 'shell10'                                  # bounded: one vox10 frame (~4 s); the vox11/12 clouds would take 15-30 s per rate / block set

    def run(name, threads):
        orc.set_threads(threads)
        c = synthetic.cloud(name).numpy()
        c4 = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
        t0 = time.perf_counter()
        enc = orc.encode(sd_np, c4)
        out = orc.decode(sd_np, enc['coords8'], enc['F'], enc['H'], enc['num_points'])
        dt = time.perf_counter() - t0
        assert len(out) == len(c4)
        return len(c4), dt

    n, dt = run(sample, cores)
    small = 'shell9' if sample in ('shell10', 'shell11', 'shell12') else 'shell7'
    n1, dt1 = run(small, 1)
    orc.set_threads(cores)
    return {'value': round(n / dt / 1e6, 5), 'unit': 'Mpoints/s', 'cores': cores, 'kind': 'port',
            'sample': f'{sample} ({n} points, one encode+decode, {dt:.1f} s; oracle C restatement, OpenMP threads = the container CPU quota; '
                      'NOT MinkowskiEngine-CPU)',
            'one_thread': {'value': round(n1 / dt1 / 1e6, 5), 'sample': f'{small} ({n1} points, one encode+decode, {dt1:.1f} s)'}}


if __name__ == '__main__':
    main()

"""The N>1 path on CPU: world_size-2 gloo processes exercise unit sharding, the 5-scalar reduction and the
variable-length coordinate gather (the data path itself has no collective)."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pcgcv2_amd import shard, synthetic


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, tmp):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        units = list(range(7))
        mine = shard.shard_units(len(units))
        assert mine == units[rank::world]
        st = shard.Stats()
        for u in mine:                                  # fake per-unit results: deterministic functions of the unit id
            st.add(bits=1000 + u, n_in=100 * (u + 1), n_out=100 * (u + 1) - u, sse_ab=0.5 * u, sse_ba=0.25 * u)
        st.reduce()
        want = np.array([sum(1000 + u for u in units), sum(100 * (u + 1) for u in units), sum(100 * (u + 1) - u for u in units),
                         sum(0.5 * u for u in units), sum(0.25 * u for u in units)])
        np.testing.assert_allclose(st.v.numpy(), want)
        s = st.summary(res=1024)
        assert abs(s['bpp'] - want[0] / want[1]) < 1e-12 and s['d1_psnr'] > 0
        rows = torch.arange((rank + 2) * 3, dtype=torch.int32).reshape(-1, 3) + 100 * rank
        got = shard.gather_varlen(rows, dst=0)
        if rank == 0:
            exp = torch.cat([torch.arange((r + 2) * 3, dtype=torch.int32).reshape(-1, 3) + 100 * r for r in range(world)])
            assert torch.equal(got, exp)
        else:
            assert got is None
        open(os.path.join(tmp, f'ok{rank}'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_reduction_and_gather(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'ok0').exists() and (tmp_path / 'ok1').exists()


def test_shard_units_single_process_and_explicit_ranks():
    assert shard.shard_units(5) == [0, 1, 2, 3, 4]
    assert [shard.shard_units(8, r, 4) for r in range(4)] == [[0, 4], [1, 5], [2, 6], [3, 7]]
    assert shard.shard_units(2, 3, 4) == []


def test_split_octants_partitions_the_cloud():
    pts = synthetic.shell('shell8').numpy()
    blocks = shard.split_octants(pts, levels=1)
    assert len(blocks) == 8
    allidx = np.sort(np.concatenate(blocks))
    np.testing.assert_array_equal(allidx, np.arange(len(pts)))
    sizes = np.array([len(b) for b in blocks])
    assert sizes.min() > 0.5 * sizes.mean() and sizes.max() < 1.6 * sizes.mean()      # balanced (lattice-aligned cuts)
    boxes = [(pts[b].min(0), pts[b].max(0)) for b in blocks]
    for i in range(8):                                   # blocks are disjoint boxes ...
        for j in range(i + 1, 8):
            assert any(boxes[i][1][d] < boxes[j][0][d] or boxes[j][1][d] < boxes[i][0][d] for d in range(3))
    cells = [set(map(tuple, (pts[b] >> 3).tolist())) for b in blocks]                    # ... that share no stride-8 cell
    assert sum(len(c) for c in cells) == len(set().union(*cells))
    off = pts + np.array([704, 8, 96])                   # position inside the cube does not matter (lattice-aligned shift)
    assert [len(b) for b in shard.split_octants(off, 1)] == [len(b) for b in blocks]
    assert len(shard.split_octants(pts[:0], 1)) == 0

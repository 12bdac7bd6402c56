"""How much of the children-level kernels' MFMA work multiplies absent neighbour-parents, and how much of that a tile ORDER
can remove (CPU probe through the oracle; no GPU).

A children-level tile = 16 parents; a halo cell's MFMAs can be skipped (wave-uniform branch) iff none of the 16 parents has the
neighbour parent the cell belongs to.  For each decoder stage of one encode+decode this prints, per parent order, the fraction
of (tile, neighbour-parent) groups that are all-absent, weighted by the halo cells (= MFMA work) each neighbour parent carries:
centre 8, face 4, edge 2, corner 1 cells.

    python tools/pattern_probe.py [shell9|shell10] [--true-geometry]
"""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import pcgc_oracle as orc           # noqa: E402  (tools/ is measurement scaffolding, not the product)
from pcgcv2_amd import synthetic                # noqa: E402

OFFS = [(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]         # k = (dz+1)*9 + (dy+1)*3 + dx+1
CELLS = np.array([(2 - abs(dx)) * (2 - abs(dy)) * (2 - abs(dz)) for dx, dy, dz in OFFS])        # halo cells per neighbour parent


def masks(coords, stride):
    """[n, 27] bool: neighbour parent k present."""
    c = (np.asarray(coords)[:, -3:] // stride).astype(np.int64)
    key = lambda a: (a[:, 0] + 2) + ((a[:, 1] + 2) << 21) + ((a[:, 2] + 2) << 42)
    keys = np.sort(key(c))
    out = np.zeros((len(c), 27), bool)
    for k, (dx, dy, dz) in enumerate(OFFS):
        q = key(c + np.array([dx, dy, dz]))
        i = np.searchsorted(keys, q)
        i[i >= len(keys)] = 0
        out[:, k] = keys[i] == q
    return out


def morton(c):
    c = c.astype(np.int64)
    m = np.zeros(len(c), np.int64)
    for b in range(12):
        for a in range(3):
            m |= ((c[:, a] >> b) & 1) << (3 * b + a)
    return m


def skip_fraction(m, order, tile=16):
    """fraction of cell work in all-absent (tile, neighbour parent) groups; and the present-row fraction of the rest."""
    m = m[order]
    n = len(m) // tile * tile
    t = m[:n].reshape(-1, tile, 27)
    any_ = t.any(1)                                            # [tiles, 27]
    work = (np.ones_like(any_) * CELLS).sum()
    kept = (any_ * CELLS).sum()
    useful = (t.sum(1) * CELLS).sum() / tile
    return 1 - kept / work, useful / kept


def pattern_key(m, sig):
    """integer key from the mask with neighbour parents ranked by `sig` (most significant first)."""
    key = np.zeros(len(m), np.int64)
    for k in sig:
        key = (key << 1) | m[:, k]
    return key


def report(name, coords, stride):
    m = masks(coords, stride)
    n = len(m)
    c = np.asarray(coords)[:, -3:] // stride
    print(f'--- {name}: {n} parents, mean present neighbour parents {m.sum(1).mean():.2f} / 27, '
          f'cell-weighted present {((m * CELLS).sum(1)).mean() / 64:.3f}, distinct masks {len(np.unique(pattern_key(m, range(27))))}')
    by_weight = sorted(range(27), key=lambda k: -CELLS[k])
    # rank by how evenly a bit splits the set (most informative first) within the weight classes
    p = m.mean(0)
    by_info = sorted(range(27), key=lambda k: (-CELLS[k], abs(p[k] - 0.5)))
    orders = {
        'as stored': np.arange(n),
        'morton': np.argsort(morton(c), kind='stable'),
        'mask (k order)': np.argsort(pattern_key(m, range(27)), kind='stable'),
        'mask (faces first)': np.argsort(pattern_key(m, by_weight), kind='stable'),
        'mask (faces first, balanced)': np.argsort(pattern_key(m, by_info), kind='stable'),
    }
    key_fw = pattern_key(m, by_weight)
    for chunk in (256, 1024, 4096, 16384):
        orders[f'stored-order chunks of {chunk}, faces-first inside'] = np.lexsort((key_fw, np.arange(n) // chunk))
    mo = np.argsort(morton(c), kind='stable')
    rank = np.empty(n, np.int64); rank[mo] = np.arange(n)
    for chunk in (256, 1024, 4096, 16384):
        orders[f'morton chunks of {chunk}, faces-first inside'] = np.lexsort((key_fw, rank // chunk))
    for bits in (7, 13, 19):
        orders[f'faces-first key, top {bits} bits only'] = np.argsort(key_fw >> (27 - bits), kind='stable')
    # greedy refinement: faces-first key, then inside runs of equal 7-bit face key sort by edge bits, ... is what the key does already.
    for label, o in orders.items():
        s, u = skip_fraction(m, o)
        print(f'{label:32s} skippable cell work {100 * s:5.1f} %   present rows in the rest {100 * u:5.1f} %')
    # bound: every tile holds one exact pattern
    print(f'{"(bound: one pattern per tile)":32s} skippable cell work {100 * (1 - (m * CELLS).sum() / (n * 64)):5.1f} %')


def main():
    name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 'shell9'
    pts = synthetic.shell(name).numpy()
    c4 = np.concatenate([np.zeros((len(pts), 1), np.int32), pts], 1)
    if '--true-geometry' in sys.argv:
        for s in (2, 4, 8):
            q = np.unique((pts // s), axis=0)
            report(f'true geometry, parents at stride {s}', q * s, s)
        return
    sd = synthetic.state_dict_to_numpy(synthetic.synthetic_state_dict())
    orc.set_threads(os.cpu_count())
    enc = orc.encode(sd, c4)
    nums = np.frombuffer(enc['num_points'], np.int32).tolist()
    shape = np.frombuffer(enc['H'][:8], np.int32)
    min_v = np.frombuffer(enc['H'][9:13], np.float32)[0]
    max_v = np.frombuffer(enc['H'][13:17], np.float32)[0]
    yF = orc.eb_decompress(orc.pack_eb_params(sd), enc['F'], min_v, max_v, shape)
    # replay decoder_forward, looking at the parent level of each stage
    C_, x, stride = enc['yC'], yF, 8
    for l in range(3):
        report(f'decoder stage {l}: parents at stride {stride} (children level: {8 * len(C_)} rows)', C_, stride)
        x = orc.relu(orc.conv_up2(x, sd[f'decoder.up{l}.kernel'], sd[f'decoder.up{l}.bias']))
        lvl = orc.Level(orc.children_coords(C_, stride), stride // 2)
        stride //= 2
        x = orc.relu(orc._conv3(sd, f'decoder.conv{l}', lvl, x))
        x = orc._block(sd, f'decoder.block{l}', lvl, x)
        cls = orc._conv3(sd, f'decoder.conv{l}_cls', lvl, x)
        mask = orc.topk_mask(cls[:, 0], nums[l])
        C_, x = lvl.C[mask], x[mask]


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Offline probe of context models for the native `_C.bin` octree coder: ideal adaptive code length (sequential
probability estimates with the coder's own update rule) of the breadth-first child-occupancy bits of a cloud's stride-8 level under
candidate contexts.  CPU only, slow Python: an evaluation tool, not the codec."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from pcgcv2_amd import synthetic

def stride8(name):
    pts = synthetic.shell(name) if name in synthetic.SHELLS else synthetic.cloud(name)
    p = np.asarray(pts.cpu().numpy() if hasattr(pts, 'cpu') else pts, np.int64) // 8
    return np.unique(p, axis=0)

def morton(p):
    def spread(v):
        v = v & 0x1FFFFF
        v = (v | v << 32) & 0x1F00000000FFFF; v = (v | v << 16) & 0x1F0000FF0000FF
        v = (v | v << 8) & 0x100F00F00F00F00F; v = (v | v << 4) & 0x10C30C30C30C30C3
        v = (v | v << 2) & 0x1249249249249249
        return v
    return spread(p[:, 0]) | (spread(p[:, 1]) << 1) | (spread(p[:, 2]) << 2)

class Adaptive:
    """the coder's estimator: 12-bit probability of a zero, p += (4096 - p) >> s on 0, p -= p >> s on 1"""
    def __init__(self, shift=4, init=None): self.p = {}; self.s = shift; self.bits = 0.0; self.init = init or {}
    def code(self, ctx, bit):
        p = self.p.get(ctx, self.init.get(ctx, 2048))
        q = p / 4096.0
        self.bits += -math.log2(q if bit == 0 else 1 - q)
        self.p[ctx] = p + ((4096 - p) >> self.s) if bit == 0 else p - (p >> self.s)

def run(pts, model, depth=None, est=None):
    depth = depth or int(np.ceil(np.log2(pts.max() + 1)))
    est = est or Adaptive()
    levels = []
    for l in range(depth + 1):
        q = np.unique(pts >> (depth - l), axis=0)
        levels.append(q)
    for l in range(depth):
        par = levels[l]; ch = levels[l + 1]
        pset = set(map(tuple, par.tolist()))
        cset = set(map(tuple, ch.tolist()))
        order = np.argsort(morton(par), kind='stable')
        pm = {tuple(par[i].tolist()): r for r, i in enumerate(order)}           # Morton rank of each parent
        coded_children = set()
        bucket = min(3, depth - 1 - l)
        for i in order:
            P = tuple(par[i].tolist())
            rank = pm[P]
            occ_before = 0
            for j in range(8):
                c = (2 * P[0] + (j & 1), 2 * P[1] + ((j >> 1) & 1), 2 * P[2] + (j >> 2))
                bit = 1 if c in cset else 0
                def nb(o):
                    """occupancy knowledge of the child-level neighbour c + o: (known at child level?, value)"""
                    cc = (c[0] + o[0], c[1] + o[1], c[2] + o[2])
                    PP = (cc[0] >> 1, cc[1] >> 1, cc[2] >> 1)
                    if PP == P:
                        jj = (cc[0] & 1) | ((cc[1] & 1) << 1) | ((cc[2] & 1) << 2)
                        return (True, cc in coded_children) if jj < j else (False, None)
                    r = pm.get(PP)
                    if r is None: return (True, False)                            # no such parent: empty for sure
                    if r < rank: return (True, cc in cset)
                    return (False, True)                                           # parent occupied, children unknown
                ctx = model(bucket, j, occ_before, nb, P, pset)
                est.code(ctx, bit)
                if bit:
                    coded_children.add(c); occ_before += 1
    return est.bits

def s_of(j): return [1 if (j >> a) & 1 else -1 for a in range(3)]

def m_current(bucket, j, before, nb, P, pset):
    s = s_of(j)
    axis = 0
    for a in range(3):
        o = [0, 0, 0]; o[a] = s[a]
        k, v = nb(tuple(o))
        axis |= (1 if v else 0) << a                                               # (unknown -> parent occupied -> 1, as the coder does)
    cnt = sum(((P[0] + d[0], P[1] + d[1], P[2] + d[2]) in pset) for d in ((1,0,0),(-1,0,0),(0,1,0),(0,-1,0),(0,0,1),(0,0,-1)))
    return (bucket, axis, j, before, (cnt + 1) // 2)

def m_octant(bucket, j, before, nb, P, pset):
    """faces as now (3-bit pattern) + the child's 4 diagonal octant neighbours (3 edges + corner) as a count, known/unknown folded in"""
    s = s_of(j)
    axis = 0
    for a in range(3):
        o = [0, 0, 0]; o[a] = s[a]
        axis |= (1 if nb(tuple(o))[1] else 0) << a
    diag = 0
    for o in ((s[0], s[1], 0), (s[0], 0, s[2]), (0, s[1], s[2]), (s[0], s[1], s[2])):
        diag += 1 if nb(o)[1] else 0
    cnt = sum(((P[0] + d[0], P[1] + d[1], P[2] + d[2]) in pset) for d in ((1,0,0),(-1,0,0),(0,1,0),(0,-1,0),(0,0,1),(0,0,-1)))
    return (bucket, axis, j, min(before, 3), (cnt + 1) // 2, min(diag, 3))

def m_inward(bucket, j, before, nb, P, pset):
    """+ what is known about the INWARD face neighbours (siblings / the opposite side): the three inward faces are siblings — known if earlier"""
    s = s_of(j)
    axis = 0
    for a in range(3):
        o = [0, 0, 0]; o[a] = s[a]
        axis |= (1 if nb(tuple(o))[1] else 0) << a
    inward = 0; known = 0
    for a in range(3):
        o = [0, 0, 0]; o[a] = -s[a]
        k, v = nb(tuple(o))
        if k and PPsame(o): pass
    return None

def PPsame(o): return True

def m_sib(bucket, j, before, nb, P, pset):
    """faces (3) + diag count (0..3+) + the pattern of the already-coded siblings that TOUCH this child by a face (up to 3 bits, by index)"""
    s = s_of(j)
    axis = 0
    for a in range(3):
        o = [0, 0, 0]; o[a] = s[a]
        axis |= (1 if nb(tuple(o))[1] else 0) << a
    sib = 0; nk = 0
    for a in range(3):
        if (j >> a) & 1:                                                            # the face sibling j ^ (1 << a) has a smaller index: known
            o = [0, 0, 0]; o[a] = -1
            sib |= (1 if nb(tuple(o))[1] else 0) << nk; nk += 1
    diag = 0
    for o in ((s[0], s[1], 0), (s[0], 0, s[2]), (0, s[1], s[2]), (s[0], s[1], s[2])):
        diag += 1 if nb(o)[1] else 0
    cnt = sum(((P[0] + d[0], P[1] + d[1], P[2] + d[2]) in pset) for d in ((1,0,0),(-1,0,0),(0,1,0),(0,-1,0),(0,0,1),(0,0,-1)))
    return (bucket, axis, j, nk, sib, min(before, 3), (cnt + 1) // 2, min(diag, 2))

def m_count(bucket, j, before, nb, P, pset):
    """geometry-agnostic: number of occupied among ALL known-or-assumed neighbours touching the child: 3 outward faces, 4 octant diagonals,
    earlier face siblings; + before; no child index except through the number of known siblings"""
    s = s_of(j)
    f = sum(1 if nb(tuple(s[a] if b == a else 0 for b in range(3)))[1] else 0 for a in range(3))
    diag = sum(1 if nb(o)[1] else 0 for o in ((s[0], s[1], 0), (s[0], 0, s[2]), (0, s[1], s[2]), (s[0], s[1], s[2])))
    sib = 0; nk = 0
    for a in range(3):
        if (j >> a) & 1:
            o = [0, 0, 0]; o[a] = -1
            sib += 1 if nb(tuple(o))[1] else 0; nk += 1
    cnt = sum(((P[0] + d[0], P[1] + d[1], P[2] + d[2]) in pset) for d in ((1,0,0),(-1,0,0),(0,1,0),(0,-1,0),(0,0,1),(0,0,-1)))
    return (bucket, f, min(diag, 3), nk, sib, min(before, 4), (cnt + 1) // 2, 7 - j if before == 0 else 0)

MODELS = {'current': m_current, 'octant': m_octant, 'sib': m_sib, 'count': m_count}
if __name__ == '__main__':
    names = sys.argv[1].split(',') if len(sys.argv) > 1 else ['shell10', 'noisy10']
    models = sys.argv[2].split(',') if len(sys.argv) > 2 else list(MODELS)
    for name in names:
        pts = stride8(name)
        for m in models:
            for sh in (4, 5):
                bits = run(pts, MODELS[m], est=Adaptive(shift=sh))
                print(f'{name:10s} {len(pts):6d} pts  {m:8s} shift {sh}: {bits / len(pts):.3f} bit / point (from p = 1/2, single stream)', flush=True)


def training_clouds():
    """integer-defined shapes in a 128^3 grid (the C++ coder can build the same ones): a sphere shell, an ellipsoid shell, a tilted plane
    slab, and the sphere with hashed drop-outs + salt"""
    g = np.arange(128)
    X, Y, Z = np.meshgrid(g, g, g, indexing='ij')
    out = []
    d2 = (X - 64) ** 2 + (Y - 64) ** 2 + (Z - 64) ** 2
    out.append(np.stack(np.nonzero((d2 >= 45 * 45) & (d2 < 46 * 46)), 1))
    e = 4 * (X - 64) ** 2 + 9 * (Y - 64) ** 2 + 16 * (Z - 64) ** 2                 # ellipsoid, semi-axes 54 / 36 / 27
    out.append(np.stack(np.nonzero((e >= 108 * 108) & (e < 112 * 112)), 1))
    pl = 3 * X + 5 * Y + 7 * Z                                                      # tilted plane through the cube
    out.append(np.stack(np.nonzero((pl >= 960) & (pl < 969) & (X > 8) & (X < 120) & (Y > 8) & (Y < 120) & (Z > 8) & (Z < 120)), 1))
    h = (X * 73856093 ^ Y * 19349663 ^ Z * 83492791) & 1023
    d3 = (X - 60) ** 2 + (Y - 66) ** 2 + (Z - 62) ** 2
    out.append(np.stack(np.nonzero((((d3 >= 38 * 38) & (d3 < 39 * 39)) & (h >= 100)) | ((h < 2) & (d3 < 50 * 50))), 1))
    return out


def evaluate(models, tests, shift=4, train=True):
    for m in models:
        est = Adaptive(shift=shift)
        if train:
            for tc in training_clouds():
                run(tc, MODELS[m], depth=7, est=est)
        prior = dict(est.p)
        for name in tests:
            pts = stride8(name)
            e2 = Adaptive(shift=shift, init=prior)
            bits = run(pts, MODELS[m], est=e2)
            print(f'{name:10s} {len(pts):6d} pts  {m:8s} shift {shift} {"trained prior" if train else "p = 1/2"}: {bits / len(pts):.3f} bit / point', flush=True)


if __name__ == '__main__' and len(sys.argv) > 3 and sys.argv[3] == 'train':
    evaluate(models, names, shift=4)


class TwoRate(Adaptive):
    """mean of a fast (shift 4) and a slow (shift 7) estimate of the same context"""
    def __init__(self, init=None): self.p = {}; self.bits = 0.0; self.init = init or {}
    def code(self, ctx, bit):
        a, b = self.p.get(ctx, self.init.get(ctx, (2048 * 16, 2048 * 16)))           # 16-bit fixed point
        q = (a + b) / 2 / 65536.0
        q = min(max(q, 1 / 4096), 1 - 1 / 4096)
        self.bits += -math.log2(q if bit == 0 else 1 - q)
        if bit == 0: a += (65536 - a) >> 4; b += (65536 - b) >> 7
        else: a -= a >> 4; b -= b >> 7
        self.p[ctx] = (a, b)


def evaluate2(models, tests, groups=8):
    for m in models:
        for kind in ('one', 'two'):
            est = Adaptive(shift=4) if kind == 'one' else TwoRate()
            for tc in training_clouds():
                run(tc, MODELS[m], depth=7, est=est)
            prior = dict(est.p)
            for name in tests:
                pts = stride8(name)
                depth = int(np.ceil(np.log2(pts.max() + 1)))
                mk = lambda: (Adaptive(shift=4, init=prior) if kind == 'one' else TwoRate(init=prior))
                e1 = mk(); single = run(pts, MODELS[m], depth=depth, est=e1)
                order = np.argsort(morton(pts), kind='stable')
                tiled = 0.0
                for g in range(groups):
                    sub = pts[order[g * len(pts) // groups:(g + 1) * len(pts) // groups]]
                    e2 = mk(); tiled += run(sub, MODELS[m], depth=depth, est=e2)
                print(f'{name:10s} {m:8s} {kind}-rate, mixed prior: single stream {single / len(pts):.3f}   {groups} independent groups {tiled / len(pts):.3f} bit / point', flush=True)


if __name__ == '__main__' and len(sys.argv) > 3 and sys.argv[3] == 'tiled':
    evaluate2(models, names)


class CountRate(Adaptive):
    """state-dependent rate: a context adapts fast on its first visits of THIS stream (shift 2, 3, then 4), starting from the prior"""
    def __init__(self, init=None, first=(2, 3, 3)): self.p = {}; self.n = {}; self.bits = 0.0; self.init = init or {}; self.first = first
    def code(self, ctx, bit):
        p = self.p.get(ctx, self.init.get(ctx, 2048))
        n = self.n.get(ctx, 0)
        s = self.first[n] if n < len(self.first) else 4
        q = p / 4096.0
        self.bits += -math.log2(q if bit == 0 else 1 - q)
        p = p + ((4096 - p) >> s) if bit == 0 else p - (p >> s)
        self.p[ctx] = min(max(p, 32), 4064); self.n[ctx] = n + 1


def evaluate3(models, tests):
    for m in models:
        est = Adaptive(shift=4)
        for tc in training_clouds():
            run(tc, MODELS[m], depth=7, est=est)
        prior = dict(est.p)
        for name in tests:
            pts = stride8(name)
            depth = int(np.ceil(np.log2(pts.max() + 1)))
            order = np.argsort(morton(pts), kind='stable')
            row = []
            for label, mk in (('shift 4', lambda: Adaptive(shift=4, init=prior)), ('fast start 2,3,3', lambda: CountRate(init=prior)),
                              ('fast start 3,3,3,3', lambda: CountRate(init=prior, first=(3, 3, 3, 3)))):
                for groups in (1, 4, 8):
                    tot = 0.0
                    for g in range(groups):
                        sub = pts[order[g * len(pts) // groups:(g + 1) * len(pts) // groups]]
                        e2 = mk(); tot += run(sub, MODELS[m], depth=depth, est=e2)
                    row.append(f'{label} x{groups}: {tot / len(pts):.3f}')
            print(f'{name:10s} {m:8s} ' + '   '.join(row), flush=True)


if __name__ == '__main__' and len(sys.argv) > 3 and sys.argv[3] == 'rates':
    evaluate3(models, names)

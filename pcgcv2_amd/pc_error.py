"""D1 (point-to-point) and D2 (point-to-plane) geometry distortion (reference pc_error.py:27-74 -> external mpeg-pcc-dmetric 0.13.4 binary).

`pc_error(infile1, infile2, res)` keeps the reference's signature and DataFrame column names.  If a `pc_error_d`
executable is installed (env PCGC_PC_ERROR or next to this file) it is invoked exactly like the reference does;
otherwise the metric is computed natively (exact nearest neighbours on the integer lattice), pinned to the binary's
output by tests/golden/d1_metric.npz (D1) and tests/golden/d2_metric.npz (D2: the reference's test.py:74-75 asks for it with
`normal=True`; normals come from infile1, as with the binary's `-n infile1`).  The metric sits outside the timed encode/decode path
(coder.py:180-182)."""
import os
import subprocess
import numpy as np
import pandas as pd

rootdir = os.path.split(__file__)[0]


def _exe():
    p = os.environ.get('PCGC_PC_ERROR') or os.path.join(rootdir, 'pc_error_d')
    return p if os.path.isfile(p) and os.access(p, os.X_OK) else None


def number_in_line(line):
    number = None
    for item in line.split(' '):
        try:
            number = float(item)
        except ValueError:
            continue
    return number


def d1_sums(a, b):
    """-> (sum of squared NN distances a->b, max) with an exact KD-tree search (scipy, host)."""
    from scipy.spatial import cKDTree
    d, _ = cKDTree(np.asarray(b, dtype=np.float64)).query(np.asarray(a, dtype=np.float64), workers=-1)
    d2 = np.rint(d * d)                      # integer lattices: squared distances are integers
    return float(d2.sum()), float(d2.max() if len(d2) else 0.0)


def d1_psnr(a, b, res):
    """mseF,PSNR (p2point) = 10*log10(3*peak^2 / max(mse1, mse2)), peak = res-1 (pc_error.py:49)."""
    s1, h1 = d1_sums(a, b)
    s2, h2 = d1_sums(b, a)
    mse1, mse2 = s1 / len(a), s2 / len(b)
    peak = float(res - 1)
    psnr = lambda m: float(10 * np.log10(3 * peak * peak / m)) if m > 0 else float('inf')
    return {'mse1      (p2point)': mse1, 'mse1,PSNR (p2point)': psnr(mse1), 'h.       1(p2point)': h1, 'h.,PSNR  1(p2point)': psnr(h1),
            'mse2      (p2point)': mse2, 'mse2,PSNR (p2point)': psnr(mse2), 'h.       2(p2point)': h2, 'h.,PSNR  2(p2point)': psnr(h2),
            'mseF      (p2point)': max(mse1, mse2), 'mseF,PSNR (p2point)': psnr(max(mse1, mse2)),
            'h.        (p2point)': max(h1, h2), 'h.,PSNR   (p2point)': psnr(max(h1, h2))}


# ---- D2 (point-to-plane), as mpeg-pcc-dmetric 0.13.4 computes it with `-n infile1` (averageNormals on, its default).  Restated from the tool's
# behaviour and pinned to its output (golden G6):
#   * normals of B ("scaleNormals"): every point of A adds its normal to its nearest neighbour(s) in B — ALL neighbours at the nearest distance, up
#     to 30 — and a point of B takes the average of what it received; a point of B that received nothing takes the average normal of its own
#     nearest neighbour(s) in A;
#   * A -> B: for a in A with nearest neighbour(s) b in B (ties as above): c2p(a) = mean over the tied b of ((a - b) . n_b)^2; mse = mean over A,
#     h. = max over A.  B -> A the same with A's own normals.  PSNR = 10 log10(3 peak^2 / mse), peak = res - 1.
_D2_TIES = 30


def _nn_ties(tree, q, chunk=1 << 15):
    """for every query point: indices of its up to 30 nearest neighbours in `tree`, squared distances, and which of them tie with the nearest"""
    k = min(_D2_TIES, tree.n)
    pts = tree.data
    for s in range(0, len(q), chunk):
        qq = q[s:s + chunk]
        _, idx = tree.query(qq, k=k, workers=-1)
        idx = idx.reshape(len(qq), k)
        e = qq[:, None, :] - pts[idx]
        d2 = (e * e).sum(-1)                                     # (exact for lattice points: ties are compared as the tool compares them)
        yield s, idx, e, d2, d2 == d2[:, :1]


def d2_estimate_normals(a, na, b):
    """normals of cloud b from those of cloud a (the tool's scaleNormals)"""
    from scipy.spatial import cKDTree
    a, b, na = np.asarray(a, np.float64), np.asarray(b, np.float64), np.asarray(na, np.float64)
    acc, cnt = np.zeros((len(b), 3)), np.zeros(len(b), np.int64)
    for s, idx, _, _, same in _nn_ties(cKDTree(b), a):
        rows, cols = np.nonzero(same)
        np.add.at(acc, idx[rows, cols], na[s + rows])
        np.add.at(cnt, idx[rows, cols], 1)
    nb = np.zeros((len(b), 3))
    got = cnt > 0
    nb[got] = acc[got] / cnt[got, None]
    lone = np.nonzero(~got)[0]
    if len(lone):
        for s, idx, _, _, same in _nn_ties(cKDTree(a), b[lone]):
            w = same.astype(np.float64)
            nb[lone[s:s + len(idx)]] = (na[idx] * w[:, :, None]).sum(1) / w.sum(1, keepdims=True)
    return nb


def d2_sums(p, q, nq):
    """p -> q with q's normals: (sum of c2c, max c2c, sum of c2p, max c2p)"""
    from scipy.spatial import cKDTree
    p, q, nq = np.asarray(p, np.float64), np.asarray(q, np.float64), np.asarray(nq, np.float64)
    s_c2c = s_c2p = 0.0
    h_c2c = h_c2p = 0.0
    for _, idx, e, d2, same in _nn_ties(cKDTree(q), p):
        proj = (e * nq[idx]).sum(-1) ** 2
        w = same.astype(np.float64)
        c2p = (proj * w).sum(1) / w.sum(1)
        s_c2c += float(d2[:, 0].sum()); s_c2p += float(c2p.sum())
        h_c2c = max(h_c2c, float(d2[:, 0].max())); h_c2p = max(h_c2p, float(c2p.max()))
    return s_c2c, h_c2c, s_c2p, h_c2p


def d2_psnr(a, na, b, res):
    """every column the reference parses from `pc_error_d -a A -b B -n A` (pc_error.py:37-47): p2point and p2plane, computed on the host"""
    nb = d2_estimate_normals(a, na, b)
    s1, h1, p1, _ = d2_sums(a, b, nb)
    s2, h2, p2, _ = d2_sums(b, a, na)
    mse1, mse2, pl1, pl2 = s1 / len(a), s2 / len(b), p1 / len(a), p2 / len(b)
    peak = float(res - 1)
    psnr = lambda m: float(10 * np.log10(3 * peak * peak / m)) if m > 0 else float('inf')
    return {'mse1      (p2point)': mse1, 'mse1,PSNR (p2point)': psnr(mse1), 'h.       1(p2point)': h1, 'h.,PSNR  1(p2point)': psnr(h1),
            'mse2      (p2point)': mse2, 'mse2,PSNR (p2point)': psnr(mse2), 'h.       2(p2point)': h2, 'h.,PSNR  2(p2point)': psnr(h2),
            'mseF      (p2point)': max(mse1, mse2), 'mseF,PSNR (p2point)': psnr(max(mse1, mse2)),
            'h.        (p2point)': max(h1, h2), 'h.,PSNR   (p2point)': psnr(max(h1, h2)),
            'mse1      (p2plane)': pl1, 'mse1,PSNR (p2plane)': psnr(pl1), 'mse2      (p2plane)': pl2, 'mse2,PSNR (p2plane)': psnr(pl2),
            'mseF      (p2plane)': max(pl1, pl2), 'mseF,PSNR (p2plane)': psnr(max(pl1, pl2))}


def ply_has_normals(path):
    """True iff the ASCII PLY's vertex element declares nx, ny and nz (header scan only)"""
    names = []
    try:
        with open(path, 'rb') as f:
            for line in f:
                t = line.decode('ascii', 'replace').split()
                if t and t[0] == 'property':
                    names.append(t[-1])
                elif t and t[0] == 'end_header':
                    break
    except OSError:
        return False
    return all(c in names for c in ('nx', 'ny', 'nz'))


def read_ply_ascii_with_normals(path):
    """ASCII PLY -> (coordinates float64 [n, 3], normals float64 [n, 3] or None): the vertex properties x y z and, when present, nx ny nz
    (read as the tool reads them: single precision)"""
    names, n, skip = [], 0, 0
    with open(path, 'rb') as f:
        in_vertex = False
        for line in f:
            skip += 1
            t = line.decode('ascii', 'replace').split()
            if not t:
                continue
            if t[0] == 'element':
                in_vertex = t[1] == 'vertex'
                if in_vertex:
                    n = int(t[2])
            elif t[0] == 'property' and in_vertex:
                names.append(t[-1])
            elif t[0] == 'end_header':
                break
    if not all(c in names for c in 'xyz'):
        raise ValueError(f'{path}: no x / y / z vertex properties')
    data = pd.read_csv(path, sep=r'\s+', header=None, skiprows=skip, nrows=n, dtype=np.float64, engine='c').to_numpy()
    xyz = data[:, [names.index(c) for c in 'xyz']]
    if not all(c in names for c in ('nx', 'ny', 'nz')):
        return xyz, None
    nrm = data[:, [names.index(c) for c in ('nx', 'ny', 'nz')]].astype(np.float32).astype(np.float64)
    return xyz, nrm


def d1_psnr_device(a, b, res, radius=12):
    """Same metrics as d1_psnr, computed on the GPU from two device coordinate tensors [N,4] (or sparse tensors' .C):
    exact nearest neighbours by ascending-distance probes of the coordinate hash (pcgc_d1_nn).  Points farther than `radius`
    voxels from the other cloud (never the case for codec outputs) are finished on the host."""
    from . import ops
    out = []
    for p, q in ((a, b), (b, a)):
        s, m, u = ops.d1_nn(p, q, radius)
        s, m, u = float(s.item()), float(m.item()), int(u.item())
        if u:                                               # rare: finish the far points exactly on the host
            pc, qc = p[:, 1:].cpu().numpy(), q[:, 1:].cpu().numpy()
            from scipy.spatial import cKDTree
            d, _ = cKDTree(qc.astype(np.float64)).query(pc.astype(np.float64), workers=-1)
            d2 = np.rint(d * d)
            s, m = float(d2.sum()), float(d2.max())
        out.append((s, m))
    (s1, h1), (s2, h2) = out
    mse1, mse2 = s1 / a.shape[0], s2 / b.shape[0]
    peak = float(res - 1)
    psnr = lambda v: float(10 * np.log10(3 * peak * peak / v)) if v > 0 else float('inf')
    return {'mse1      (p2point)': mse1, 'mse1,PSNR (p2point)': psnr(mse1), 'h.       1(p2point)': h1, 'h.,PSNR  1(p2point)': psnr(h1),
            'mse2      (p2point)': mse2, 'mse2,PSNR (p2point)': psnr(mse2), 'h.       2(p2point)': h2, 'h.,PSNR  2(p2point)': psnr(h2),
            'mseF      (p2point)': max(mse1, mse2), 'mseF,PSNR (p2point)': psnr(max(mse1, mse2)),
            'h.        (p2point)': max(h1, h2), 'h.,PSNR   (p2point)': psnr(max(h1, h2)),
            'sse1': s1, 'sse2': s2}


def pc_error(infile1, infile2, res, normal=False, show=False):
    exe = _exe()
    if exe is None:
        from .data_utils import read_ply_ascii_geo
        if normal:
            a, na = read_ply_ascii_with_normals(infile1)
            if na is None:
                raise ValueError(f'{infile1} has no normals (nx ny nz): point-to-plane (D2) needs them, as `pc_error_d -n` does')
            return pd.DataFrame([d2_psnr(a, na, read_ply_ascii_with_normals(infile2)[0], res)])
        return pd.DataFrame([d1_psnr(read_ply_ascii_geo(infile1), read_ply_ascii_geo(infile2), res)])
    headers = ['mse1      (p2point)', 'mse1,PSNR (p2point)', 'h.       1(p2point)', 'h.,PSNR  1(p2point)',
               'mse2      (p2point)', 'mse2,PSNR (p2point)', 'h.       2(p2point)', 'h.,PSNR  2(p2point)',
               'mseF      (p2point)', 'mseF,PSNR (p2point)', 'h.        (p2point)', 'h.,PSNR   (p2point)']
    p2plane = ['mse1      (p2plane)', 'mse1,PSNR (p2plane)', 'mse2      (p2plane)', 'mse2,PSNR (p2plane)',
               'mseF      (p2plane)', 'mseF,PSNR (p2plane)']
    cmd = [exe, '-a', infile1, '-b', infile2, '--hausdorff=1', '--resolution=' + str(res - 1)]
    if normal:
        headers += p2plane
        cmd += ['-n', infile1]
    out = subprocess.run(cmd, stdout=subprocess.PIPE).stdout.decode('utf-8', 'replace')
    results = {}
    for line in out.splitlines():
        if show:
            print(line)
        for key in headers:
            if line.find(key) != -1:
                results[key] = number_in_line(line)
    return pd.DataFrame([results])

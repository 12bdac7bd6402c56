"""D1 (point-to-point) geometry distortion (reference pc_error.py:27-74 -> external mpeg-pcc-dmetric 0.13.4 binary).

`pc_error(infile1, infile2, res)` keeps the reference's signature and DataFrame column names.  If a `pc_error_d`
executable is installed (env PCGC_PC_ERROR or next to this file) it is invoked exactly like the reference does;
otherwise the metric is computed natively (exact nearest neighbours on the integer lattice), pinned to the binary's
output by tests/golden/d1_metric.npz.  The metric sits outside the timed encode/decode path (coder.py:180-182)."""
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
            raise NotImplementedError('point-to-plane (D2) needs normals and the external pc_error_d binary')
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

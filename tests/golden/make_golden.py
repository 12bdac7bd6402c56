#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference implementation.

Runs ONLY in the authoring container (needs /root/reference).  The reference's Python
never travels: this script imports the importable slices of it (with empty stub modules
for the un-installed third-party packages MinkowskiEngine / torchac / h5py), feeds them
seeded inputs and stores inputs + outputs as small .npz / .txt data files.

Fixtures:
  G1 entropy_tables.npz   EntropyBottleneck params -> _likelihood / pmf / cdf   (entropy_model.py:82-149)
  G2 ordering.npz         array2vector keys / argsort, istopk masks            (data_utils.py:55-89)
  G3 ply_format.npz       bytes written by write_ply_ascii_geo + read-back     (data_utils.py:19-48)
  G4 d1_metric.npz        pc_error_d (mpeg-pcc-dmetric 0.13.4) outputs         (pc_error.py:27-74)
  G6 d2_metric.npz        the same binary with `-n infile1` (normal=True): p2plane columns (pc_error.py:40-47,51-53; test.py:74-75)
  G5 state_dict_keys.txt  EntropyBottleneck state_dict names / shapes          (entropy_model.py:59-80)
"""
import os, sys, types, shutil, subprocess, tempfile
import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))

for name in ('torchac', 'h5py', 'MinkowskiEngine'):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, REF)
import entropy_model as ref_em      # noqa: E402
import data_utils as ref_du         # noqa: E402
import pc_error as ref_pe           # noqa: E402


def pack_params(eb):
    """Flatten the 12 unique parameter tensors in a fixed order: matrices 0..3, biases 0..3, factors 0..3."""
    parts = []
    for lst in (eb._matrices, eb._biases, eb._factors):
        for p in lst:
            parts.append(p.detach().numpy().astype(np.float32).ravel())
    return np.concatenate(parts)


def g1():
    cases = {}
    # (seed, perturb, lo, hi): alphabets from 1 to 300 symbols — torch's CPU bmm switches implementation with the size
    # (a plain loop below ~44 symbols, the BLAS batched GEMM above), and vectorised / scalar-tail transcendental paths
    # alternate with the element count, so the cases cover small, medium and large L with odd and even lengths
    G1_CASES = [(1234, False, -8, 9), (7, True, -20, 20), (99, True, -3, 2), (5, True, 0, 0), (11, True, -40, 37),
                (21, True, -75, 74), (22, True, -150, 149), (23, True, -128, 127), (24, True, -17, 25), (25, True, -21, 22),
                (26, True, -22, 22), (27, True, -31, 32), (28, True, -100, 56), (29, False, -60, 61), (30, True, -7, 7)]
    for ci, (seed, perturb, lo, hi) in enumerate(G1_CASES):
        np.random.seed(seed); torch.manual_seed(seed)
        eb = ref_em.EntropyBottleneck(8)
        if perturb:
            with torch.no_grad():
                for f in eb._factors:
                    f.copy_(torch.empty_like(f).uniform_(-0.5, 0.5))
                for m in eb._matrices:
                    m.add_(torch.empty_like(m).uniform_(-0.3, 0.3))
        min_v = torch.tensor(float(lo)); max_v = torch.tensor(float(hi))
        symbols = torch.arange(min_v, max_v + 1).reshape(-1, 1).repeat(1, 8)
        with torch.no_grad():
            lik = eb._likelihood(symbols)                           # [L,8]
            pmf = torch.clamp(lik, min=eb._likelihood_bound).permute(1, 0)
            cdf = eb._pmf_to_cdf(pmf)                               # [8,L+1]
        cases[f'c{ci}_params'] = pack_params(eb)
        cases[f'c{ci}_minmax'] = np.array([lo, hi], np.float32)
        cases[f'c{ci}_likelihood'] = lik.numpy()
        cases[f'c{ci}_pmf'] = pmf.contiguous().numpy()
        cases[f'c{ci}_cdf'] = cdf.contiguous().numpy()
        if ci == 0:
            with open(os.path.join(OUT, 'state_dict_keys.txt'), 'w') as f:
                for k, v in eb.state_dict().items():
                    f.write(f'{k} {list(v.shape)}\n')
    cases['n_cases'] = np.array(len(G1_CASES))
    cases['cpu_capability'] = np.array(torch.backends.cpu.get_cpu_capability())
    cases['torch_version'] = np.array(torch.__version__)
    cases['cpu_vendor'] = np.array([l.split(':')[1].strip() for l in open('/proc/cpuinfo') if l.startswith('vendor_id')][0])
    np.savez_compressed(os.path.join(OUT, 'entropy_tables.npz'), **cases)


class _Duck:
    """duck-typed stand-in for the ME.SparseTensor attributes istopk touches (data_utils.py:77-89)."""
    def __init__(self, F):
        self.F = F; self.device = F.device
        self._batchwise_row_indices = [torch.arange(len(F))]
    def __len__(self): return len(self.F)


def g2():
    rng = np.random.default_rng(2024)
    out = {}
    for i, (n, hi) in enumerate([(257, 128), (1000, 1024), (5000, 64)]):
        c = rng.integers(0, hi, size=(n, 3)).astype(np.int32)
        c = np.unique(c, axis=0); rng.shuffle(c)
        c4 = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
        t = torch.tensor(c4)
        key = ref_du.array2vector(t, t.max() + 1)
        out[f's{i}_coords'] = c4
        out[f's{i}_key'] = key.numpy()
        out[f's{i}_argsort'] = np.argsort(key.numpy())
    for i, (n, k) in enumerate([(64, 10), (1000, 391), (4096, 4096), (333, 1)]):
        v = rng.permutation(n).astype(np.float32) * 0.37 - 50.0      # tie-free
        m = ref_du.istopk(_Duck(torch.tensor(v).reshape(-1, 1)), [k])
        out[f't{i}_vals'] = v; out[f't{i}_k'] = np.array(k); out[f't{i}_mask'] = m.numpy()
    np.savez_compressed(os.path.join(OUT, 'ordering.npz'), **out)


def g3():
    rng = np.random.default_rng(3)
    c = rng.integers(0, 1024, size=(37, 3)).astype(np.int64)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, 'a.ply')
        ref_du.write_ply_ascii_geo(p, c)
        raw = open(p, 'rb').read()
        back = ref_du.read_ply_ascii_geo(p)
        # a PLY with extra header lines and float-formatted coords, as 8iVFB files have
        p2 = os.path.join(d, 'b.ply')
        with open(p2, 'w') as f:
            f.write('ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 3\nproperty float x\n'
                    'property float y\nproperty float z\nproperty uchar red\nend_header\n'
                    '1.000 2.000 3.000 255\n10 20 30 0\n7.0 8.0 9.0 1\n')
        raw2 = open(p2, 'rb').read()
        back2 = ref_du.read_ply_ascii_geo(p2)
    np.savez_compressed(os.path.join(OUT, 'ply_format.npz'), coords=c, file_bytes=np.frombuffer(raw, np.uint8),
                        read_back=back, file2_bytes=np.frombuffer(raw2, np.uint8), read_back2=back2)


def g4():
    rng = np.random.default_rng(4)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, 'pc_error_d'); shutil.copy(os.path.join(REF, 'pc_error_d'), exe); os.chmod(exe, 0o755)
        ref_pe.rootdir = d
        for i, (n, res, drop, jit) in enumerate([(2000, 64, 1, 0.3), (5000, 1024, 50, 0.5), (3000, 1024, 0, 0.0),
                                                 (4000, 256, 400, 0.8)]):
            a = np.unique(rng.integers(0, res, size=(n, 3)), axis=0)
            b = a.copy()
            if drop: b = np.delete(b, rng.choice(len(b), drop, replace=False), 0)
            mv = rng.random(len(b)) < jit
            b[mv] = np.clip(b[mv] + rng.integers(-2, 3, size=(mv.sum(), 3)), 0, res - 1)
            b = np.unique(b, axis=0)
            pa, pb = os.path.join(d, f'a{i}.ply'), os.path.join(d, f'b{i}.ply')
            ref_du.write_ply_ascii_geo(pa, a); ref_du.write_ply_ascii_geo(pb, b)
            df = ref_pe.pc_error(pa, pb, res=res)
            out[f'p{i}_a'] = a.astype(np.int32); out[f'p{i}_b'] = b.astype(np.int32); out[f'p{i}_res'] = np.array(res)
            for key in ('mse1      (p2point)', 'mse2      (p2point)', 'mseF      (p2point)', 'mseF,PSNR (p2point)',
                        'mse1,PSNR (p2point)', 'mse2,PSNR (p2point)', 'h.        (p2point)'):
                out[f'p{i}_' + key.replace(' ', '').replace(',', '_')] = np.array(df[key][0])
    out['n_cases'] = np.array(4)
    np.savez_compressed(os.path.join(OUT, 'd1_metric.npz'), **out)


def g6():
    """point-to-plane columns: clouds with normals through the vendored binary, invoked by the imported reference pc_error(normal=True)"""
    rng = np.random.default_rng(6)
    out = {}

    def write_with_normals(path, pts, nrm):
        with open(path, 'w') as f:
            f.write('ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n'
                    'property float nx\nproperty float ny\nproperty float nz\nend_header\n' % len(pts))
            for q, m in zip(pts, nrm):
                f.write('%d %d %d %.6f %.6f %.6f\n' % (q[0], q[1], q[2], m[0], m[1], m[2]))

    def shell(res, radius, thick):
        g = np.stack(np.meshgrid(*[np.arange(res)] * 3, indexing='ij'), -1).reshape(-1, 3)
        c = np.array([res / 2.0] * 3)
        r = np.linalg.norm(g - c, axis=1)
        a = g[np.abs(r - radius) < thick]
        return a, (a - c) / np.linalg.norm(a - c, axis=1, keepdims=True)

    cases = []
    a, na = shell(64, 20, 0.7)                                   # thin sphere shell, analytic normals; B: jittered by one voxel, 50 points fewer
    cases.append((a, na, np.unique(np.clip(a + rng.integers(-1, 2, size=a.shape), 0, 63), axis=0)[:len(a) - 50], 64))
    a, na = shell(128, 44, 1.2)                                  # thicker shell; B: shifted up to three voxels (many exact distance ties)
    cases.append((a, na, np.unique(np.clip(a + rng.integers(-3, 4, size=a.shape), 0, 127), axis=0), 128))
    a, na = shell(64, 18, 0.6)
    cases.append((a, na, a.copy(), 64))                          # identical clouds: zeros, infinite PSNR
    a = np.unique(rng.integers(0, 128, size=(4000, 3)), axis=0)  # scattered points, random unit normals, B sparser than A
    na = rng.standard_normal((len(a), 3)); na /= np.linalg.norm(na, axis=1, keepdims=True)
    b = np.unique(np.clip(a + rng.integers(-2, 3, size=a.shape), 0, 127), axis=0)
    cases.append((a, na, b[rng.random(len(b)) < 0.6], 128))
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, 'pc_error_d'); shutil.copy(os.path.join(REF, 'pc_error_d'), exe); os.chmod(exe, 0o755)
        ref_pe.rootdir = d
        for i, (a, na, b, res) in enumerate(cases):
            pa, pb = os.path.join(d, f'a{i}.ply'), os.path.join(d, f'b{i}.ply')
            write_with_normals(pa, a, na); ref_du.write_ply_ascii_geo(pb, b)
            df = ref_pe.pc_error(pa, pb, res=res, normal=True)
            out[f'p{i}_a'] = a.astype(np.int32); out[f'p{i}_b'] = b.astype(np.int32); out[f'p{i}_res'] = np.array(res)
            out[f'p{i}_na'] = np.array([[float('%.6f' % v) for v in row] for row in na], np.float32)      # (as the file holds them)
            for key in df.columns:
                out[f'p{i}_' + key.replace(' ', '').replace(',', '_')] = np.array(df[key][0])
    out['n_cases'] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, 'd2_metric.npz'), **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g6']
    for name in which:
        {'g1': g1, 'g2': g2, 'g3': g3, 'g4': g4, 'g6': g6}[name]()
    print('golden fixtures written to', OUT)

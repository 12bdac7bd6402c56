"""Deterministic synthetic inputs and weights: the reference's test PLYs and checkpoints are external
downloads that are not available where this code runs, so benchmarks and parity tests use

  * `shell(...)`: a perturbed-sphere voxel surface.  On a G^3 grid with voxel centres p = (i,j,k) - (G-1)/2, a voxel is
    occupied iff | |p| - (R + A*sin(a*theta)*cos(b*phi)) | < 0.5, theta = acos(p_z/|p|), phi = atan2(p_y, p_x).
    `shell10` (G=1024, R=250, A=18, lobes (3,5)) has 786 632 points — the stand-in for longdress_vox10_1300.ply.
  * `synthetic_state_dict(...)`: seeded weights in the reference's {'model': state_dict} layout.
"""
import math
import numpy as np
import torch

SHELLS = {
    # name: (grid, radius, amplitude, (lobe_theta, lobe_phi))
    'shell10': (1024, 250.0, 18.0, (3, 5)),
    'shell10_b': (1024, 243.0, 18.0, (2, 7)),
    'shell10_c': (1024, 236.0, 18.0, (4, 3)),
    'shell10_d': (1024, 262.0, 18.0, (5, 4)),
    'shell11': (2048, 455.0, 33.0, (3, 5)),
    'shell12': (4096, 620.0, 45.0, (3, 5)),
    'shell9': (512, 125.0, 9.0, (3, 5)),
    'shell8': (256, 62.0, 4.5, (3, 5)),
    'shell7': (128, 31.0, 2.2, (3, 5)),
    'shell6': (64, 15.0, 1.1, (3, 5)),
}


def shell(name='shell10', device='cpu'):
    """-> int32 [N,3] voxel coordinates in x-fastest raster order (torch tensor on `device`)."""
    return _shell(*SHELLS[name], device=device)


def _shell(grid, radius, amp, lobes, device='cpu'):
    la, lb = lobes
    dev = torch.device(device)
    c = (grid - 1) / 2.0
    lo = max(0, int(math.floor(c - radius - amp - 2)))
    hi = min(grid, int(math.ceil(c + radius + amp + 3)))
    ax = torch.arange(lo, hi, dtype=torch.float64, device=dev) - c
    out = []
    X, Y = ax.view(1, -1), ax.view(-1, 1)                    # one z-slab at a time: [y, x]
    rxy2 = X * X + Y * Y
    phi = torch.atan2(Y, X).expand(len(ax), len(ax))
    cosb = torch.cos(lb * phi)
    for zi in range(lo, hi):
        pz = zi - c
        r = torch.sqrt(rxy2 + pz * pz)
        theta = torch.acos(torch.clamp(pz / r, -1.0, 1.0))
        target = radius + amp * torch.sin(la * theta) * cosb
        m = (r - target).abs() < 0.5
        if m.any():
            yx = m.nonzero()
            out.append(torch.stack([yx[:, 1] + lo, yx[:, 0] + lo, torch.full_like(yx[:, 0], zi)], 1))
    return torch.cat(out, 0).to(torch.int32)


# ---- geometry that is NOT a smooth closed surface ---------------------------------------------------------------------------------
# The reference publishes results on seven different clouds (results/*.csv: human bodies, a dancer, a building, a statue); every
# `shell` above is one family — closed, smooth, genus 0, about two voxels thick, rows in raster order.  The families below cover what
# that family never produces: full 27-neighbourhoods and masses of EXACTLY equal logits (solid: the interior of a filled body sees the
# same all-ones neighbourhood everywhere, so whole regions tie in the top-k selection and the tie rule decides which voxels survive),
# isolated voxels / ragged rims / holes (noisy), several components that intersect, touch or float apart, one-voxel sheets and rods
# (multi), a thinned surface whose decoder has to produce several children per parent (sparse: the up-sampling case rho > 1).
# Every cloud is a pure function of its name (numpy Generator with a fixed seed), so tests and the oracle see the same rows.
CLOUDS = {
    # name: (family, grid scale s) — sizes scale with s so that the CPU suite can run small versions of the same shapes
    'solid_cube': ('solid_cube', 1.0),       # 80^3 = 512 000 voxels
    'solid_ball': ('solid_ball', 1.0),       # radius 50.5: ~539 000 voxels
    'noisy10': ('noisy', 1.0),               # perturbed shell (vox10) with holes, 10 % drop-outs, near-surface and volume salt: ~0.73 M
    'multi10': ('multi', 1.0),               # two intersecting shells + flat and tilted sheets + rod + solid block + far-away shell: ~0.9 M
    'sparse10': ('sparse', 1.0),             # vox10 shell thinned to ~0.5 M points (the rho = 4 operating point of coder.py:107)
    'solid_cube_s': ('solid_cube', 0.3), 'solid_ball_s': ('solid_ball', 0.3), 'noisy_s': ('noisy', 0.125), 'multi_s': ('multi', 0.125),
    'sparse_s': ('sparse', 0.125),
}


def _shell_np(center, radius, half_thickness=0.5):
    """voxels (x, y, z) whose centre lies within half_thickness of the sphere |p - center| = radius -> int64 [n, 3]"""
    c = np.asarray(center, np.float64)
    lo = np.floor(c - radius - half_thickness - 1).astype(np.int64)
    hi = np.ceil(c + radius + half_thickness + 2).astype(np.int64)
    out = []
    xs, ys = np.arange(lo[0], hi[0]), np.arange(lo[1], hi[1])
    X, Y = np.meshgrid(xs - c[0], ys - c[1], indexing='xy')                      # [y, x]
    rxy2 = X * X + Y * Y
    for z in range(lo[2], hi[2]):
        m = np.abs(np.sqrt(rxy2 + (z - c[2]) ** 2) - radius) < half_thickness
        if m.any():
            yi, xi = np.nonzero(m)
            out.append(np.stack([xs[xi], ys[yi], np.full(len(xi), z)], 1))
    return np.concatenate(out, 0) if out else np.zeros((0, 3), np.int64)


def _grid3(lo, hi):
    ax = [np.arange(l, h) for l, h in zip(lo, hi)]
    g = np.stack(np.meshgrid(*ax, indexing='ij'), -1).reshape(-1, 3)
    return g.astype(np.int64)


def _canonical(points, grid):
    """unique voxels inside [0, grid)^3 in x-fastest raster order (z major) -> int64 [n, 3]"""
    p = points[((points >= 0) & (points < grid)).all(1)]
    key = (p[:, 2] << 40) | (p[:, 1] << 20) | p[:, 0]
    key = np.unique(key)
    return np.stack([key & 0xFFFFF, (key >> 20) & 0xFFFFF, key >> 40], 1)


def cloud(name, order='raster', seed=0, device='cpu'):
    """-> int32 [N, 3] voxel coordinates of a named cloud: any `shell*` of SHELLS or any entry of CLOUDS.  order: 'raster' (x fastest,
    z major — how the shells come) or 'shuffled' (the same rows under the permutation numpy's default_rng(seed) draws: a PLY written by
    a scanner or a mesh sampler has no particular row order, and the canonical row order of every level here follows the INPUT order)."""
    if name in SHELLS:
        pts = shell(name).numpy().astype(np.int64)
    else:
        family, s = CLOUDS[name]
        rng = np.random.default_rng(20260928)
        if family == 'solid_cube':
            e = max(4, int(round(80 * s)))
            grid = 128
            pts = _grid3((24, 24, 24), (24 + e, 24 + e, 24 + e))
        elif family == 'solid_ball':
            grid, r = 128, 50.5 * s
            g = _grid3((0, 0, 0), (grid, grid, grid)) if s >= 1 else _grid3((32, 32, 32), (96, 96, 96))
            pts = g[((g - 63.5) ** 2).sum(1) < r * r]
        elif family in ('noisy', 'sparse'):
            grid = max(64, int(1024 * s))
            c = (grid - 1) / 2.0
            R, A = 243.0 * s, 18.0 * s
            # a perturbed shell like shell10_b (the same generator, so that it scales with s)
            g_lo, g_hi = int(max(0, np.floor(c - R - A - 2))), int(min(grid, np.ceil(c + R + A + 3)))
            surf = _shell(grid, R, A, (2, 7)).numpy().astype(np.int64)
            if family == 'sparse':
                pts = surf[rng.random(len(surf)) >= 0.34]                      # ~0.5 M of 0.76 M voxels: most stride-2 parents survive
            else:
                holes = surf[rng.integers(0, len(surf), 40)]
                radii = rng.uniform(10 * s, 30 * s, 40)
                keep = np.ones(len(surf), bool)
                for h, hr in zip(holes, radii):
                    keep &= ((surf - h) ** 2).sum(1) > hr * hr
                keep &= rng.random(len(surf)) >= 0.10                           # drop-outs
                near = surf[rng.integers(0, len(surf), int(60000 * s * s))] + rng.integers(-3, 4, (int(60000 * s * s), 3))
                salt = rng.integers(g_lo, g_hi, (int(40000 * s * s), 3))        # isolated voxels anywhere in the bounding cube
                pts = np.concatenate([surf[keep], near, salt], 0)
        elif family == 'multi':
            grid = max(64, int(1024 * s))
            u = lambda *v: tuple(x * s for x in v)
            sheet = _grid3((int(230 * s), int(230 * s), int(300 * s)), (int(790 * s), int(790 * s), int(300 * s) + 1))
            sheet = sheet[((sheet[:, :2] - 511.5 * s) ** 2).sum(1) < (280 * s) ** 2]                 # flat one-voxel disc
            t = _grid3((int(150 * s), int(150 * s), 0), (int(650 * s), int(650 * s), 1))
            tilt = np.stack([t[:, 0], t[:, 1], (t[:, 0] * 3 + t[:, 1]) // 5 + int(380 * s)], 1)      # tilted sheet: stair steps, gaps in z
            tt = np.arange(int(800 * s))
            rod = np.stack([int(100 * s) + tt, int(100 * s) + tt, int(100 * s) + (tt * 7) // 8], 1)  # one-voxel diagonal rod
            block = _grid3((int(800 * s),) * 3, (int(800 * s) + max(3, int(24 * s)),) * 3)             # small solid block
            pts = np.concatenate([_shell_np(u(390, 512, 560), 140 * s), _shell_np(u(600, 512, 560), 140 * s),     # two intersecting shells
                                  sheet, tilt, rod, block, _shell_np(u(880, 140, 880), 40 * s)], 0)               # + one far away
        else:
            raise KeyError(name)
        pts = _canonical(pts, grid)
    if order == 'shuffled':
        pts = pts[np.random.default_rng(seed).permutation(len(pts))]
    elif order != 'raster':
        raise ValueError("order must be 'raster' or 'shuffled'")
    return torch.from_numpy(np.ascontiguousarray(pts, dtype=np.int32)).to(device)


def synthetic_state_dict(seed=1234, gain=50.0):
    """Reference-layout state_dict with seeded weights: conv kernel ~ U(-a,a), a = 1/sqrt(K*Cin); bias ~ U(-0.1,0.1);
    entropy parameters as entropy_model.py:66-80 (np.random.seed) then _factors ~ U(-0.5,0.5);
    encoder.conv3 (the latent-producing conv) scaled by `gain` so that round(y.F) spans a realistic alphabet."""
    from .pcc_model import PCCModel
    np.random.seed(seed)
    g = torch.Generator().manual_seed(seed)
    model = PCCModel()
    sd = model.state_dict()
    out = {}
    for k, v in sd.items():
        if k.startswith('entropy_bottleneck'):
            continue
        if k.endswith('.kernel'):
            vol_cin = v.shape[0] if v.dim() == 2 else v.shape[0] * v.shape[1]
            a = 1.0 / math.sqrt(vol_cin)
            out[k] = (torch.rand(v.shape, generator=g) * 2 - 1) * a
        else:
            out[k] = (torch.rand(v.shape, generator=g) * 2 - 1) * 0.1
    out['encoder.conv3.kernel'] = out['encoder.conv3.kernel'] * gain
    out['encoder.conv3.bias'] = out['encoder.conv3.bias'] * gain
    eb = model.entropy_bottleneck
    for i in range(4):
        out[f'entropy_bottleneck._matrices.{i}'] = eb._matrices[i].detach().clone()
        out[f'entropy_bottleneck._biases.{i}'] = eb._biases[i].detach().clone()
        out[f'entropy_bottleneck._factors.{i}'] = (torch.rand(eb._factors[i].shape, generator=g) - 0.5)
    out['entropy_bottleneck.matrix'] = out['entropy_bottleneck._matrices.3']
    out['entropy_bottleneck.bias'] = out['entropy_bottleneck._biases.3']
    out['entropy_bottleneck.factor'] = out['entropy_bottleneck._factors.3']
    return {k: v.float().contiguous() for k, v in out.items()}


def state_dict_to_numpy(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}

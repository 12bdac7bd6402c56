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
    grid, radius, amp, (la, lb) = SHELLS[name]
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

#!/usr/bin/env python3
"""Which torch-CPU operator makes the reference's CDF table host-dependent?  Prints, per golden case, a checksum after every
operator of the reference-order evaluation (entropy_model.py:82-149) on THIS host; run it on two hosts and diff the output.
Also compares the final fp32 cdf with golden G1 (generated on the authoring host)."""
import hashlib, os, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pcgc_oracle as orc


def h(t):
    return hashlib.md5(t.contiguous().numpy().tobytes()).hexdigest()[:8]


def main():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'entropy_tables.npz'))
    print('host:', torch.backends.cpu.get_cpu_capability(), [l.split(':')[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')][0],
          'torch', torch.__version__, 'mkl', torch.backends.mkl.is_available(), 'golden from', g['cpu_capability'])
    for ci in range(int(g['n_cases'])):
        M, B, Fa = orc._eb_unpack(g[f'c{ci}_params'])
        lo, hi = g[f'c{ci}_minmax']
        x = torch.arange(float(lo), float(hi) + 1).reshape(-1, 1).repeat(1, 8).permute(1, 0).contiguous()
        shape = x.size(); x = x.view(shape[0], 1, -1)
        out = [f'c{ci} L={int(hi - lo) + 1}']
        ends = []
        for half in (-0.5, 0.5):
            z = x + half
            for i in range(4):
                sp = F.softplus(M[i]); out.append('sp' + h(sp))
                z = torch.matmul(sp, z); out.append('mm' + h(z))
                z += B[i]
                th = torch.tanh(z); out.append('th' + h(th))
                z += torch.tanh(Fa[i]) * th; out.append('z' + h(z))
            ends.append(z)
        lower, upper = ends
        sign = -torch.sign(torch.add(lower, upper))
        su, sl = torch.sigmoid(sign * upper), torch.sigmoid(sign * lower)
        out.append('sg' + h(su) + h(sl))
        lik = torch.abs(su - sl).view(shape).permute(1, 0)
        pmf = torch.clamp(lik, min=1e-9).permute(1, 0)
        cdf = pmf.cumsum(dim=-1); out.append('cs' + h(cdf))
        cdf = torch.cat([torch.zeros(pmf.shape[:-1] + (1,)), cdf], dim=-1).clamp(max=1.)
        gold = g[f'c{ci}_cdf']
        nd = int((cdf.numpy() != gold).sum())
        nq = int((orc.cdf_u16(cdf.numpy()) != orc.cdf_u16(gold)).sum())
        out.append(f'cdf_vs_golden: {nd} fp32 / {nq} u16 of {gold.size} differ')
        print(' '.join(out))


if __name__ == '__main__':
    main()

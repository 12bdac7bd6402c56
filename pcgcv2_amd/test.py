"""Rate-distortion sweep (the role of the reference's `test.py`): one cloud through a list of checkpoints, one CSV row per
rate, files namespaced by the per-rate postfix `_r{i}` (test.py:38).  `test(...)` keeps the reference's signature and CSV
column names; `sweep(...)` is the underlying generator.

What differs in execution: the input tensor — and its optional down-scaled version — is built once, and because the
encoder's geometry pyramid and kernel maps depend only on the coordinates they are built by the first rate and reused by
all others (they are cached on the tensor's coordinate levels).  D2 (point-to-plane) columns need normals in the
input PLY (as the reference's `pc_error(..., normal=True)` does) and are computed natively when no `pc_error_d` binary is installed.  Checkpoints may be paths or in-memory state dicts."""
import os
import time

import numpy as np
import pandas as pd
import torch

from .coder import Coder, stream_bits
from .data_utils import load_sparse_tensor, scale_sparse_tensor, write_ply_ascii_geo
from .pc_error import pc_error, ply_has_normals
from .pcc_model import PCCModel

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')

REFERENCE_CKPTS = ['./ckpts/r1_0.025bpp.pth', './ckpts/r2_0.05bpp.pth', './ckpts/r3_0.10bpp.pth', './ckpts/r4_0.15bpp.pth',
                   './ckpts/r5_0.25bpp.pth', './ckpts/r6_0.3bpp.pth', './ckpts/r7_0.4bpp.pth']


def _state_dict(ckpt):
    if isinstance(ckpt, (str, os.PathLike)):
        if not os.path.exists(ckpt):
            raise FileNotFoundError(ckpt)
        return torch.load(ckpt, map_location=device)['model']
    return ckpt


def _timed(fn):
    torch.cuda.synchronize()
    t0 = time.time()
    out = fn()
    torch.cuda.synchronize()
    return out, round(time.time() - t0, 3)


def sweep(filedir, ckpts, outdir, scaling_factor=1.0, rho=1.0, res=1024):
    """Yield one single-row DataFrame per checkpoint (columns as in the reference's results/*.csv)."""
    x = load_sparse_tensor(filedir, device)
    os.makedirs(outdir, exist_ok=True)
    prefix = os.path.join(outdir, os.path.split(filedir)[-1].split('.')[0])
    x_in = scale_sparse_tensor(x, factor=scaling_factor) if scaling_factor != 1 else x
    model = PCCModel().to(device)
    with_normals = ply_has_normals(filedir)       # (test.py:74-75 always asks for D2: its test clouds carry normals; a cloud without them gets D1 only)
    for rate, ckpt in enumerate(ckpts, start=1):
        model.load_state_dict(_state_dict(ckpt))
        coder = Coder(model=model, filename=prefix)
        tag = f'_r{rate}'
        _, t_enc = _timed(lambda: coder.encode(x_in, postfix=tag))
        x_dec, t_dec = _timed(lambda: coder.decode(postfix=tag, rho=rho))
        if scaling_factor != 1:
            x_dec = scale_sparse_tensor(x_dec, factor=1.0 / scaling_factor)
        bits = stream_bits(prefix, tag)
        bpps = (bits / len(x)).round(3)
        dec_ply = prefix + tag + '_dec.ply'
        write_ply_ascii_geo(dec_ply, x_dec.C.detach().cpu().numpy()[:, 1:])
        row = pc_error(filedir, dec_ply, res=res, normal=with_normals, show=False)
        row["num_points(input)"], row["num_points(output)"], row["resolution"] = len(x), len(x_dec), res
        row["bits"], row["bpp"] = sum(bits).round(3), sum(bpps).round(3)
        row["bpp(coords)"], row["bpp(feats)"] = bpps[0], bpps[1]
        row["time(enc)"], row["time(dec)"] = t_enc, t_dec
        yield row


def test(filedir, ckptdir_list, outdir, resultdir, scaling_factor=1.0, rho=1.0, res=1024, verbose=True):
    """Reference entry point (test.py:13): runs the sweep, rewrites `<resultdir>/<cloud>.csv` after every rate."""
    os.makedirs(resultdir, exist_ok=True)
    csv_name = os.path.join(resultdir, os.path.split(filedir)[-1].split('.')[0] + '.csv')
    rows, table = [], None
    for rate, row in enumerate(sweep(filedir, ckptdir_list, outdir, scaling_factor, rho, res), start=1):
        rows.append(row)
        table = pd.concat(rows, ignore_index=True)
        table.to_csv(csv_name, index=False)
        if verbose:
            print(f'[r{rate}] bpp {row["bpp"][0]}  D1 {row["mseF,PSNR (p2point)"][0]:.4f} dB  enc {row["time(enc)"][0]} s  '
                  f'dec {row["time(dec)"][0]} s  -> {csv_name}')
    return table


def plot_rd(table, title, path):
    """R-D curve like test.py:123-136 (optional: needs matplotlib)."""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    fig, _ = plt.subplots(figsize=(7, 4))
    curves = [("mseF,PSNR (p2point)", "D1", 'red'), ("mseF,PSNR (p2plane)", "D2", 'blue')]
    for col, label, colour in curves:
        if col in table:
            plt.plot(np.array(table["bpp"]), np.array(table[col]), label=label, marker='x', color=colour)
    plt.title(title); plt.xlabel('bpp'); plt.ylabel('PSNR'); plt.grid(ls='-.'); plt.legend(loc='lower right')
    fig.savefig(path)


def main(argv=None):
    import argparse
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--filedir", default='../../../testdata/8iVFB/longdress_vox10_1300.ply')
    parser.add_argument("--outdir", default='./output')
    parser.add_argument("--resultdir", default='./results')
    parser.add_argument("--scaling_factor", type=float, default=1.0, help='scaling_factor')
    parser.add_argument("--res", type=int, default=1024, help='resolution')
    parser.add_argument("--rho", type=float, default=1.0, help='the ratio of the number of output points to the number of input points')
    parser.add_argument("--ckpts", nargs='*', default=REFERENCE_CKPTS)
    args = parser.parse_args(argv)
    table = test(args.filedir, args.ckpts, args.outdir, args.resultdir, scaling_factor=args.scaling_factor, rho=args.rho, res=args.res)
    name = os.path.split(args.filedir)[-1][:-4]
    try:
        plot_rd(table, name, os.path.join(args.resultdir, name + '.jpg'))
    except ImportError:
        pass


if __name__ == '__main__':
    main()

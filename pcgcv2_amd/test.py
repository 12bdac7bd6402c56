"""R-D sweep harness (reference test.py:13-136): one cloud through a list of checkpoints, one CSV row per rate.

Same function signature, same CSV column names and per-rate `_r{i}` postfixes.  Differences that matter for speed, not
for results: the input tensor is loaded once and its geometry pyramid / kernel maps (rate-independent — the encoder's
levels depend only on the coordinates) are built by the first rate and reused by the others (they are cached on the
tensor's coordinate levels); point-to-plane (D2) columns need normals and the external `pc_error_d` binary and are
omitted when it is absent.  `ckptdir_list` entries may be checkpoint paths or in-memory state dicts."""
import os
import time
import numpy as np
import pandas as pd
import torch

from .pcc_model import PCCModel
from .coder import Coder
from .data_utils import load_sparse_tensor, scale_sparse_tensor, write_ply_ascii_geo
from .pc_error import pc_error, _exe

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')


def test(filedir, ckptdir_list, outdir, resultdir, scaling_factor=1.0, rho=1.0, res=1024, verbose=True):
    log = print if verbose else (lambda *a, **k: None)
    start_time = time.time()
    x = load_sparse_tensor(filedir, device)
    log('Loading Time:\t', round(time.time() - start_time, 4), 's')
    os.makedirs(outdir, exist_ok=True)
    os.makedirs(resultdir, exist_ok=True)
    stem = os.path.split(filedir)[-1].split('.')[0]
    filename = os.path.join(outdir, stem)
    log('output filename:\t', filename)
    model = PCCModel().to(device)
    x_in = scale_sparse_tensor(x, factor=scaling_factor) if scaling_factor != 1 else x      # once: rate-independent
    rows = []
    for idx, ckpt in enumerate(ckptdir_list):
        log('=' * 10, idx + 1, '=' * 10)
        if isinstance(ckpt, (str, os.PathLike)):
            assert os.path.exists(ckpt)
            sd = torch.load(ckpt, map_location=device)['model']
            log('load checkpoint from \t', ckpt)
        else:
            sd = ckpt
        model.load_state_dict(sd)
        coder = Coder(model=model, filename=filename)
        postfix_idx = '_r' + str(idx + 1)

        torch.cuda.synchronize(); start_time = time.time()
        _ = coder.encode(x_in, postfix=postfix_idx)
        torch.cuda.synchronize(); time_enc = round(time.time() - start_time, 3)
        log('Enc Time:\t', time_enc, 's')
        start_time = time.time()
        x_dec = coder.decode(postfix=postfix_idx, rho=rho)
        torch.cuda.synchronize(); time_dec = round(time.time() - start_time, 3)
        log('Dec Time:\t', time_dec, 's')
        if scaling_factor != 1:
            x_dec = scale_sparse_tensor(x_dec, factor=1.0 / scaling_factor)

        bits = np.array([os.path.getsize(filename + postfix_idx + p) * 8 for p in ['_C.bin', '_F.bin', '_H.bin', '_num_points.bin']])
        bpps = (bits / len(x)).round(3)
        log('bits:\t', sum(bits), '\nbpps:\t', sum(bpps).round(3))

        dec_ply = filename + postfix_idx + '_dec.ply'
        write_ply_ascii_geo(dec_ply, x_dec.C.detach().cpu().numpy()[:, 1:])
        results = pc_error(filedir, dec_ply, res=res, normal=_exe() is not None, show=False)
        log('D1 PSNR:\t', results["mseF,PSNR (p2point)"][0])
        results["num_points(input)"] = len(x)
        results["num_points(output)"] = len(x_dec)
        results["resolution"] = res
        results["bits"] = sum(bits).round(3)
        results["bpp"] = sum(bpps).round(3)
        results["bpp(coords)"] = bpps[0]
        results["bpp(feats)"] = bpps[1]
        results["time(enc)"] = time_enc
        results["time(dec)"] = time_dec
        rows.append(results)
        all_results = pd.concat(rows, ignore_index=True)          # DataFrame.append (test.py:94) no longer exists in pandas 2
        csv_name = os.path.join(resultdir, stem + '.csv')
        all_results.to_csv(csv_name, index=False)
        log('Wrile results to: \t', csv_name)
    return all_results


def main(argv=None):
    import argparse
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--filedir", default='../../../testdata/8iVFB/longdress_vox10_1300.ply')
    parser.add_argument("--outdir", default='./output')
    parser.add_argument("--resultdir", default='./results')
    parser.add_argument("--scaling_factor", type=float, default=1.0, help='scaling_factor')
    parser.add_argument("--res", type=int, default=1024, help='resolution')
    parser.add_argument("--rho", type=float, default=1.0,
                        help='the ratio of the number of output points to the number of input points')
    parser.add_argument("--ckpts", nargs='*', default=['./ckpts/r1_0.025bpp.pth', './ckpts/r2_0.05bpp.pth', './ckpts/r3_0.10bpp.pth',
                                                       './ckpts/r4_0.15bpp.pth', './ckpts/r5_0.25bpp.pth', './ckpts/r6_0.3bpp.pth',
                                                       './ckpts/r7_0.4bpp.pth'])
    args = parser.parse_args(argv)
    all_results = test(args.filedir, args.ckpts, args.outdir, args.resultdir, scaling_factor=args.scaling_factor, rho=args.rho,
                       res=args.res)
    try:                                                            # R-D plot (test.py:123-136), optional
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
        fig, ax = plt.subplots(figsize=(7, 4))
        plt.plot(np.array(all_results["bpp"][:]), np.array(all_results["mseF,PSNR (p2point)"][:]), label="D1", marker='x', color='red')
        if "mseF,PSNR (p2plane)" in all_results:
            plt.plot(np.array(all_results["bpp"][:]), np.array(all_results["mseF,PSNR (p2plane)"][:]), label="D2", marker='x', color='blue')
        name = os.path.split(args.filedir)[-1][:-4]
        plt.title(name); plt.xlabel('bpp'); plt.ylabel('PSNR'); plt.grid(ls='-.'); plt.legend(loc='lower right')
        fig.savefig(os.path.join(args.resultdir, name + '.jpg'))
    except ImportError:
        pass


if __name__ == '__main__':
    main()

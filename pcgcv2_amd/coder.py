"""Codec API (reference coder.py:16-184): CoordinateCoder / FeatureCoder / Coder and the CLI, on the HIP operator set.

Same constructor / method signatures, same four files per coded cloud:
   <prefix><postfix>_C.bin           coordinates of the stride-8 latent (tmc3 stream, or native "PCGO" stream)
   <prefix><postfix>_F.bin           range-coded latent features (torchac-compatible stream)
   <prefix><postfix>_H.bin           int32[2] shape | int8 len(min_v)=1 | float32 min_v | float32 max_v   (17 bytes)
   <prefix><postfix>_num_points.bin  int32[3] = [N4, N2, N1]
"""
import os
import time
import numpy as np
import torch

from . import gpcc
from .data_utils import (array2vector, istopk, sort_spare_tensor, load_sparse_tensor, scale_sparse_tensor,
                         write_ply_ascii_geo, read_ply_ascii_geo)
from .pc_error import pc_error
from .pcc_model import PCCModel
from concurrent.futures import ThreadPoolExecutor
from .sparse import SparseTensor, CoordMap, require_gpu
from . import ops

_POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix='pcgc-coord')

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')


class CoordinateCoder():
    """coder.py:16-36.  Uses tmc3 when installed (identical temp-PLY + subprocess protocol), else the native codec."""

    def __init__(self, filename):
        self.filename = filename
        self.ply_filename = filename + '.ply'

    def encode(self, coords, postfix=''):
        coords = (coords.numpy() if isinstance(coords, torch.Tensor) else np.asarray(coords)).astype('int')
        bin_path = self.filename + postfix + '_C.bin'
        if gpcc.tmc3_path() is not None:
            write_ply_ascii_geo(filedir=self.ply_filename, coords=coords)
            gpcc.gpcc_encode(self.ply_filename, bin_path)
            os.remove(self.ply_filename)
        else:
            gpcc.native_encode(coords, bin_path)
        return

    def decode(self, postfix=''):
        bin_path = self.filename + postfix + '_C.bin'
        if gpcc.is_native_stream(bin_path):
            return gpcc.native_decode(bin_path)
        gpcc.gpcc_decode(bin_path, self.ply_filename)
        coords = read_ply_ascii_geo(self.ply_filename)
        os.remove(self.ply_filename)
        return coords


class FeatureCoder():
    """coder.py:39-70."""

    def __init__(self, filename, entropy_model):
        self.filename = filename
        self.entropy_model = entropy_model.cpu()      # no-op here: the tables are evaluated on the GPU

    def encode(self, feats, postfix=''):
        strings, min_v, max_v = self.entropy_model.compress(feats)
        shape = feats.shape
        with open(self.filename + postfix + '_F.bin', 'wb') as fout:
            fout.write(strings)
        with open(self.filename + postfix + '_H.bin', 'wb') as fout:
            fout.write(np.array(shape, dtype=np.int32).tobytes())
            fout.write(np.array(len(min_v), dtype=np.int8).tobytes())
            fout.write(np.array(min_v, dtype=np.float32).tobytes())
            fout.write(np.array(max_v, dtype=np.float32).tobytes())
        return

    def decode(self, postfix='', device=None):
        with open(self.filename + postfix + '_F.bin', 'rb') as fin:
            strings = fin.read()
        with open(self.filename + postfix + '_H.bin', 'rb') as fin:
            shape = np.frombuffer(fin.read(4 * 2), dtype=np.int32)
            len_min_v = np.frombuffer(fin.read(1), dtype=np.int8)[0]
            min_v = np.frombuffer(fin.read(4 * len_min_v), dtype=np.float32)[0]
            max_v = np.frombuffer(fin.read(4 * len_min_v), dtype=np.float32)[0]
        return self.entropy_model.decompress(strings, min_v, max_v, shape, channels=shape[-1], device=device)


class Coder():
    """coder.py:73-112."""

    def __init__(self, model, filename):
        self.model = model
        self.filename = filename
        self.coordinate_coder = CoordinateCoder(filename)
        self.feature_coder = FeatureCoder(self.filename, model.entropy_bottleneck)

    @torch.no_grad()
    def encode(self, x, postfix=''):
        """coder.py:80-91.  Same outputs, different schedule: the geometry pyramid (N1 -> N2 -> N4 -> N8) is built first, so
        the stride-8 coordinates — all the coordinate coder needs — are on the host before the convolutions are even
        enqueued, and the (sequential, host-side) coordinate coding overlaps the GPU's encoder pass."""
        c8 = x.cmap
        for _ in range(3):
            c8 = c8.down()[0]                                   # cached on the levels: the encoder reuses these maps
        perm = ops.sort_zyx(c8.C)
        y_C = ops.gather_coords(c8.C, perm)
        coords8 = y_C.cpu().numpy()[:, 1:] // c8.stride         # tiny D2H (N8 x 16 B)
        y_list = self.model.encoder(x)                          # asynchronous: ~60 kernel launches
        self.coordinate_coder.encode(coords8, postfix=postfix)  # runs on the host while the GPU computes
        y = SparseTensor(ops.gather_feats(y_list[0].F, perm), coordinate_map=CoordMap(y_C, c8.stride, unique=True))
        num_points = [len(ground_truth) for ground_truth in y_list[1:] + [x]]
        with open(self.filename + postfix + '_num_points.bin', 'wb') as f:
            f.write(np.array(num_points, dtype=np.int32).tobytes())
        self.feature_coder.encode(y.F, postfix=postfix)
        return y

    @torch.no_grad()
    def decode(self, rho=1, postfix=''):
        dev = require_gpu(next(self.model.decoder.parameters()).device)
        # the two bitstreams are independent: decode the coordinates on a helper thread while this thread range-decodes
        # the features (both are native calls that release the GIL)
        fut_C = _POOL.submit(self.coordinate_coder.decode, postfix)
        y_F = self.feature_coder.decode(postfix=postfix, device=dev)
        y_C = fut_C.result()
        # coder.py:96-99: prepend the batch column and sort with array2vector on the host; here the (tiny) list goes to the
        # device in one copy and is sorted there (same (z,y,x,batch) order)
        y_C4 = np.zeros((len(y_C), 4), dtype=np.int32)
        y_C4[:, 1:] = np.asarray(y_C, dtype=np.int32) * 8
        y_C = torch.from_numpy(y_C4).to(dev)
        y_C = ops.gather_coords(y_C, ops.sort_zyx(y_C))
        y = SparseTensor(features=y_F, coordinates=y_C, tensor_stride=8, device=dev, assume_unique=True)
        with open(self.filename + postfix + '_num_points.bin', 'rb') as fin:
            num_points = np.frombuffer(fin.read(4 * 3), dtype=np.int32).tolist()
            num_points[-1] = int(rho * num_points[-1])
            num_points = [[num] for num in num_points]
        _, out = self.model.decoder(y, nums_list=num_points, ground_truth_list=[None] * 3, training=False)
        return out


def main(argv=None):
    import argparse
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--ckptdir", default='ckpts/r3_0.10bpp.pth')
    parser.add_argument("--filedir", default='../../../testdata/8iVFB/longdress_vox10_1300.ply')
    parser.add_argument("--scaling_factor", type=float, default=1.0, help='scaling_factor')
    parser.add_argument("--rho", type=float, default=1.0,
                        help='the ratio of the number of output points to the number of input points')
    parser.add_argument("--res", type=int, default=1024, help='resolution')
    parser.add_argument("--outdir", default='./output')
    args = parser.parse_args(argv)
    filedir = args.filedir

    start_time = time.time()
    x = load_sparse_tensor(filedir, device)
    print('Loading Time:\t', round(time.time() - start_time, 4), 's')

    os.makedirs(args.outdir, exist_ok=True)
    filename = os.path.join(args.outdir, os.path.split(filedir)[-1].split('.')[0])
    print(filename)

    print('=' * 10, 'Test', '=' * 10)
    model = PCCModel().to(device)
    assert os.path.exists(args.ckptdir)
    ckpt = torch.load(args.ckptdir, map_location=device)
    model.load_state_dict(ckpt['model'])
    print('load checkpoint from \t', args.ckptdir)

    coder = Coder(model=model, filename=filename)
    x_in = scale_sparse_tensor(x, factor=args.scaling_factor) if args.scaling_factor != 1 else x

    torch.cuda.synchronize(); start_time = time.time()
    _ = coder.encode(x_in)
    torch.cuda.synchronize(); print('Enc Time:\t', round(time.time() - start_time, 3), 's')

    start_time = time.time()
    x_dec = coder.decode(rho=args.rho)
    torch.cuda.synchronize(); print('Dec Time:\t', round(time.time() - start_time, 3), 's')

    if args.scaling_factor != 1:
        x_dec = scale_sparse_tensor(x_dec, factor=1.0 / args.scaling_factor)

    bits = np.array([os.path.getsize(filename + postfix) * 8 for postfix in ['_C.bin', '_F.bin', '_H.bin', '_num_points.bin']])
    bpps = (bits / len(x)).round(3)
    print('bits:\t', bits, '\nbpps:\t', bpps)
    print('bits:\t', sum(bits), '\nbpps:\t', sum(bpps).round(3))

    start_time = time.time()
    write_ply_ascii_geo(filename + '_dec.ply', x_dec.C.detach().cpu().numpy()[:, 1:])
    print('Write PC Time:\t', round(time.time() - start_time, 3), 's')

    start_time = time.time()
    pc_error_metrics = pc_error(args.filedir, filename + '_dec.ply', res=args.res, show=False)
    print('PC Error Metric Time:\t', round(time.time() - start_time, 3), 's')
    print('D1 PSNR:\t', pc_error_metrics["mseF,PSNR (p2point)"][0])


if __name__ == '__main__':
    main()

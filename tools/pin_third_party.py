#!/usr/bin/env python3
"""Pin the oracle's restatement of MinkowskiEngine / torchac to the real libraries — ONE command for someone who has them.

The encode/decode path's arithmetic lives in two un-vendored third-party libraries: MinkowskiEngine >= 0.5
(sparse conv / generative transpose / pruning: autoencoder.py:13-50,71-136,155-237,247) and torchac 0.9.3 (16-bit CDF
normalisation + range coder: entropy_model.py:174,192).  Neither can be installed where this repository is built (no network),
so `oracle/` restates their published semantics and says "parity unpinned" for them.  Run this script in an environment that
has both (README.md:19-20 of the reference lists them) and it writes

    tests/golden/third_party.npz

— seeded inputs and the libraries' own outputs.  `tests/test_oracle_golden.py::test_oracle_matches_third_party_vectors`
(CPU) and `tests/test_gpu_parity.py::test_hip_matches_third_party_vectors` (GPU) pick the file up when it exists and compare
the oracle / the HIP path with it; without the file they skip.  Commit the .npz (it is data: inputs + expected outputs).

Two kinds of cases per operator:
  *_exact   small-integer features and weights: every product and partial sum is exactly representable in fp32, so the result
            does not depend on the summation order — any mismatch is a SEMANTIC difference (kernel-offset order, the origin of
            even kernels, which child a transposed-conv offset produces, pruning order).  Compared bit for bit.
  *_float   random fp32 data: ME sums per-offset GEMMs (cuBLAS / MKL order), this implementation one fmaf chain per output
            (DESIGN.md §3); compared within 1e-5 relative and the count of bit-identical values is reported.
Rows are matched by COORDINATE (ME's output row order for new coordinate maps is an implementation detail of its hash map);
the order of rows ME itself returns is stored too, so the canonical-order conventions (DESIGN.md §3 ‡) can be checked.

    python tools/pin_third_party.py [--out tests/golden/third_party.npz] [--device cpu|cuda]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
SEED = 20240917


def cloud(rng, n, extent):
    """unique random voxels, int32 [n, 4] (batch 0), in a fixed (shuffled) row order"""
    c = np.unique(rng.integers(0, extent, size=(4 * n, 3)), axis=0)
    c = c[rng.permutation(len(c))[:n]].astype(np.int32)
    return np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)


def surface(rng, r):
    """a voxelised sphere shell (the neighbourhood statistics of real geometry), shuffled rows"""
    g = np.arange(-r - 2, r + 3)
    x, y, z = np.meshgrid(g, g, g, indexing='ij')
    m = np.abs(np.sqrt(x * x + y * y + z * z) - r) < 0.5
    c = np.stack([x[m], y[m], z[m]], 1).astype(np.int32) + r + 2
    c = c[rng.permutation(len(c))]
    return np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)


def data(rng, shape, exact):
    if exact:
        return rng.integers(-3, 4, size=shape).astype(np.float32)
    return rng.standard_normal(shape).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden', 'third_party.npz'))
    ap.add_argument('--device', default='cpu')
    args = ap.parse_args()

    import torch
    import MinkowskiEngine as ME                     # the import line a machine without the libraries stops at
    import torchac

    dev = torch.device(args.device)
    rng = np.random.default_rng(SEED)
    out = {'me_version': np.array(ME.__version__), 'torchac_version': np.array(getattr(torchac, '__version__', 'unknown')),
           'torch_version': np.array(torch.__version__), 'device': np.array(str(dev))}
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def sparse(C, F, stride=1):
        return ME.SparseTensor(features=T(F), coordinates=T(C).int(), tensor_stride=stride, device=dev)

    def put(name, C_in, F_in, W, b, y):
        out[name + '/C_in'], out[name + '/F_in'], out[name + '/W'], out[name + '/b'] = C_in, F_in, W, b
        out[name + '/C_out'] = y.C.cpu().numpy().astype(np.int32)
        out[name + '/F_out'] = y.F.detach().cpu().numpy().astype(np.float32)
        out[name + '/stride_out'] = np.array(y.tensor_stride[0])

    clouds = {'rand': cloud(rng, 3000, 24), 'shell': surface(rng, 14)}
    for cname, C in clouds.items():
        for exact in (True, False):
            tag = f'{cname}_{"exact" if exact else "float"}'
            # ---- k3 stride 1 (autoencoder.py:13-19), k1 (:28-34), k2 stride 2 (:79-85) ----
            for k, s, cin, cout in ((3, 1, 4, 8), (3, 1, 16, 16), (1, 1, 8, 4), (2, 2, 4, 8)):
                conv = ME.MinkowskiConvolution(in_channels=cin, out_channels=cout, kernel_size=k, stride=s, bias=True, dimension=3).to(dev)
                W = data(rng, tuple(conv.kernel.shape), exact)
                b = data(rng, tuple(conv.bias.shape), exact)
                with torch.no_grad():
                    conv.kernel.copy_(T(W)); conv.bias.copy_(T(b))
                F = data(rng, (len(C), cin), exact)
                put(f'conv_k{k}s{s}_{cin}_{cout}/{tag}', C, F, W, b, conv(sparse(C, F)))
            # ---- generative transpose k2 s2 (autoencoder.py:155-161) on a stride-2 level ----
            C2 = np.unique(np.concatenate([C[:, :1], C[:, 1:] // 2 * 2], 1), axis=0).astype(np.int32)
            C2 = C2[rng.permutation(len(C2))]
            up = ME.MinkowskiGenerativeConvolutionTranspose(in_channels=4, out_channels=8, kernel_size=2, stride=2, bias=True, dimension=3).to(dev)
            W = data(rng, tuple(up.kernel.shape), exact)
            b = data(rng, tuple(up.bias.shape), exact)
            with torch.no_grad():
                up.kernel.copy_(T(W)); up.bias.copy_(T(b))
            F = data(rng, (len(C2), 4), exact)
            x2 = sparse(C2, F, stride=2)
            y = up(x2)
            put(f'up_k2s2_4_8/{tag}', C2, F, W, b, y)
            # ---- a k3 conv ON the generated level (the decoder's conv0/1/2: autoencoder.py:162-168) and pruning (:237,247) ----
            conv = ME.MinkowskiConvolution(in_channels=8, out_channels=1, kernel_size=3, stride=1, bias=True, dimension=3).to(dev)
            Wc = data(rng, tuple(conv.kernel.shape), exact)
            bc = data(rng, tuple(conv.bias.shape), exact)
            with torch.no_grad():
                conv.kernel.copy_(T(Wc)); conv.bias.copy_(T(bc))
            cls = conv(y)
            out[f'cls_on_up/{tag}/W'], out[f'cls_on_up/{tag}/b'] = Wc, bc
            out[f'cls_on_up/{tag}/C_out'] = cls.C.cpu().numpy().astype(np.int32)
            out[f'cls_on_up/{tag}/F_out'] = cls.F.detach().cpu().numpy().astype(np.float32)
            keep = rng.random(len(y)) < 0.4
            pruned = ME.MinkowskiPruning()(y, T(keep))
            out[f'prune/{tag}/mask'] = keep
            out[f'prune/{tag}/C_out'] = pruned.C.cpu().numpy().astype(np.int32)
            out[f'prune/{tag}/F_out'] = pruned.F.detach().cpu().numpy().astype(np.float32)
        # ---- ME.SparseTensor construction with duplicate coordinates (data_utils.py:108: default quantization mode) ----
        dup = np.concatenate([C[:500], C[:200], C[100:300]], 0)
        Fd = np.arange(len(dup), dtype=np.float32).reshape(-1, 1)
        try:
            xd = sparse(dup, Fd)
            out[f'dedup/{cname}/C_in'], out[f'dedup/{cname}/F_in'] = dup, Fd
            out[f'dedup/{cname}/C_out'] = xd.C.cpu().numpy().astype(np.int32)
            out[f'dedup/{cname}/F_out'] = xd.F.cpu().numpy().astype(np.float32)
        except Exception as e:                                      # some ME builds refuse duplicates in the default mode
            out[f'dedup/{cname}/error'] = np.array(repr(e))

    # ---- torchac: 16-bit normalisation + range coder (entropy_model.py:173-174,191-192) ----
    for case, (n, channels, L) in {'small': (50, 8, 5), 'latent': (4000, 8, 41), 'wide': (300, 8, 300), 'one_symbol': (64, 8, 1)}.items():
        pmf = rng.random((channels, L)).astype(np.float32) ** 3 + np.float32(1e-9)
        pmf /= pmf.sum(1, keepdims=True)
        cdf = np.concatenate([np.zeros((channels, 1), np.float32), np.cumsum(pmf, 1, dtype=np.float32)], 1).clip(max=1.0).astype(np.float32)
        # symbols drawn from the pmf (plus the extremes so every table row is exercised)
        sym = np.stack([rng.choice(L, size=n, p=pmf[c].astype(np.float64) / pmf[c].astype(np.float64).sum()) for c in range(channels)], 1).astype(np.int16)
        sym[0, :] = 0
        sym[-1, :] = L - 1
        out_cdf = torch.from_numpy(cdf).unsqueeze(0).repeat(n, 1, 1)
        data_bytes = torchac.encode_float_cdf(out_cdf, torch.from_numpy(sym), check_input_bounds=True)
        back = torchac.decode_float_cdf(out_cdf, data_bytes)
        assert np.array_equal(back.numpy().astype(np.int16), sym), 'torchac round trip failed'
        out[f'torchac/{case}/cdf'], out[f'torchac/{case}/sym'] = cdf, sym
        out[f'torchac/{case}/bytes'] = np.frombuffer(data_bytes, np.uint8)
        try:                                                         # the normalised table itself, if this torchac exposes it
            q = torchac.torchac._convert_to_int_and_normalize(out_cdf[:1], True)
            out[f'torchac/{case}/cdf_int16'] = q[0].numpy()
        except Exception:
            pass

    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, **out)
    print(f'wrote {args.out}: {len(out)} arrays (MinkowskiEngine {ME.__version__}, torch {torch.__version__})')
    print('now run:  python -m pytest tests/test_oracle_golden.py -k third_party   (and -m gpu -k third_party on an MI355X)')


if __name__ == '__main__':
    sys.exit(main())

"""Consumer of tests/golden/third_party.npz (written by tools/pin_third_party.py in an environment that has MinkowskiEngine
and torchac): replays every recorded case through a backend — the CPU oracle or the HIP path — and compares with what the
libraries themselves returned.  Rows are matched by coordinate; `_exact` cases (small-integer data: order-independent sums)
bit for bit, `_float` cases within 1e-5."""
import os
import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'third_party.npz')


def available():
    return os.path.exists(PATH)


def _by_coord(C, F):
    order = np.lexsort(C.T[::-1])
    return C[order], F[order]


def _compare(name, got_C, got_F, want_C, want_F, exact):
    gC, gF = _by_coord(np.asarray(got_C, np.int32), np.asarray(got_F, np.float32))
    wC, wF = _by_coord(np.asarray(want_C, np.int32), np.asarray(want_F, np.float32))
    np.testing.assert_array_equal(gC, wC, err_msg=f'{name}: output coordinate set differs from the library')
    if exact:
        np.testing.assert_array_equal(gF, wF, err_msg=f'{name}: semantic difference (order-independent data)')
    else:
        np.testing.assert_allclose(gF, wF, rtol=1e-5, atol=1e-5, err_msg=name)
    return float((gF == wF).mean())


def run(backend):
    """backend: object with
         conv(C_in [n,4], F_in, W, b, kernel_size, stride) -> (C_out, F_out)
         up(C_in, F_in, W, b, stride_in)                   -> (C_out, F_out)        (generative transpose k2 s2)
         prune(C, F, mask)                                 -> (C_out, F_out)
         dedup(C, F)                                       -> (C_out, F_out)
         rc_encode(cdf_float [C,L+1], sym int16 [n,C])     -> bytes
         cdf_u16(cdf_float)                                -> uint16 [C, L+1]
       -> dict case -> fraction of bit-identical values (float cases)"""
    g = np.load(PATH)
    names = sorted({k.rsplit('/', 1)[0] for k in g.files if '/' in k})
    report = {}
    for name in names:
        get = lambda f: g[f'{name}/{f}']
        exact = name.endswith('_exact')
        head = name.split('/')[0]
        if head.startswith('conv_k'):
            k, s = int(head[6]), int(head[8])
            C, F = backend.conv(get('C_in'), get('F_in'), get('W'), get('b'), k, s)
            report[name] = _compare(name, C, F, get('C_out'), get('F_out'), exact)
        elif head.startswith('up_k2s2'):
            C, F = backend.up(get('C_in'), get('F_in'), get('W'), get('b'), 2)
            report[name] = _compare(name, C, F, get('C_out'), get('F_out'), exact)
            # the decoder's next two operators on the generated level, in the library's own row order of that level
            tag = name.split('/', 1)[1]
            yC, yF = g[f'{name}/C_out'], g[f'{name}/F_out']
            cC, cF = backend.conv(yC, yF, g[f'cls_on_up/{tag}/W'], g[f'cls_on_up/{tag}/b'], 3, 1)
            report[f'cls_on_up/{tag}'] = _compare(f'cls_on_up/{tag}', cC, cF, g[f'cls_on_up/{tag}/C_out'], g[f'cls_on_up/{tag}/F_out'], exact)
            pC, pF = backend.prune(yC, yF, g[f'prune/{tag}/mask'])
            np.testing.assert_array_equal(pC, g[f'prune/{tag}/C_out'], err_msg=f'prune/{tag}: MinkowskiPruning row ORDER differs')
            np.testing.assert_array_equal(pF, g[f'prune/{tag}/F_out'])
        elif head == 'dedup' and f'{name}/C_out' in g.files:
            C, F = backend.dedup(get('C_in'), get('F_in'))
            wC, wF = _by_coord(get('C_out'), get('F_out'))
            gC, gF = _by_coord(C, F)
            np.testing.assert_array_equal(gC, wC, err_msg=f'{name}: dedup coordinate set')
            report[name] = float((gF == wF).mean())               # which duplicate ME keeps is a ‡ convention: reported, and asserted:
            assert report[name] == 1.0, f'{name}: ME keeps a different duplicate (switch conventions dedup_keep)'
        elif head == 'torchac':
            cdf, sym = get('cdf'), get('sym')
            if f'{name}/cdf_int16' in g.files:
                np.testing.assert_array_equal(backend.cdf_u16(cdf), g[f'{name}/cdf_int16'].view(np.uint16), err_msg=f'{name}: 16-bit normalisation')
            assert backend.rc_encode(cdf, sym) == g[f'{name}/bytes'].tobytes(), f'{name}: range-coder bytes differ from torchac'
            report[name] = 1.0
    return report

"""Switchable ‡ conventions (hard part 1).

Four behaviours of the reference's un-vendored dependencies decide its results at the bit level but are pinned by nothing in
/root/reference (no ME / torchac source, no test vectors).  Each is ONE switch here, with the same switch in the oracle, so that the
day a real MinkowskiEngine run disagrees the convention can be flipped instead of rewritten:

  kernel_offset_order  'xyz' (default): offset index k = (dx+1) + 3 (dy+1) + 9 (dz+1), x fastest (ME's kernel_region iteration as
                       restated in SURVEY §8a a7);  'zyx': z fastest.  Implemented as a permutation of the `kernel` tensors'
                       offset axis when a state dict is loaded (PCCModel.load_state_dict) — the kernels themselves are unchanged.
  topk_tie             'low' (default): among logits equal to the top-k threshold the lower row index is kept; 'high': the higher.
  dedup_keep           'first' (default): ME.SparseTensor construction keeps the first of several rows with equal coordinates; 'last'.
  even_kernel_origin   'floor' only: a k=2,s=2 kernel covers offsets {0,1} from floor(c / 2s) * 2s.  No alternative is implemented: any
                       other origin would change which voxels exist at the coarser strides, i.e. the bitstream's point counts.

Defaults reproduce every committed fixture and test.  Settings are process-wide; set them before building tensors / loading weights.
"""
import numpy as np

_STATE = {'kernel_offset_order': 'xyz', 'topk_tie': 'low', 'dedup_keep': 'first', 'even_kernel_origin': 'floor'}
_ALLOWED = {'kernel_offset_order': ('xyz', 'zyx'), 'topk_tie': ('low', 'high'), 'dedup_keep': ('first', 'last'), 'even_kernel_origin': ('floor',)}


def get(name):
    return _STATE[name]


def set_convention(name, value):
    if name not in _ALLOWED or value not in _ALLOWED[name]:
        raise ValueError(f'convention {name!r} must be one of {_ALLOWED.get(name)}')
    _STATE[name] = value
    if name == 'topk_tie':
        from ._lib import lib, check
        check(lib().pcgc_set_convention(0, 1 if value == 'high' else 0), 'set_convention')


def reset():
    for k, v in (('kernel_offset_order', 'xyz'), ('topk_tie', 'low'), ('dedup_keep', 'first')):
        set_convention(k, v)


def offset_permutation(volume):
    """perm with  kernel_here[k] = kernel_checkpoint[perm[k]]  for a kernel of `volume` offsets (27: k3, 8: k2), or None."""
    if _STATE['kernel_offset_order'] == 'xyz' or volume not in (27, 8):
        return None
    n = 3 if volume == 27 else 2
    perm = np.empty(volume, np.int64)
    for k in range(volume):
        x, y, z = k % n, (k // n) % n, k // (n * n)          # our index: x fastest
        perm[k] = z + n * y + n * n * x                      # the checkpoint's index: z fastest
    return perm


def permute_state_dict(sd):
    """Apply the kernel-offset convention to a reference-layout state dict (copies only the tensors it changes)."""
    if _STATE['kernel_offset_order'] == 'xyz':
        return sd
    import torch
    out = dict(sd)
    for k, v in sd.items():
        if k.endswith('.kernel') and v.dim() == 3:
            perm = offset_permutation(v.shape[0])
            if perm is not None:
                out[k] = v[torch.as_tensor(perm, device=v.device)].contiguous()
    return out

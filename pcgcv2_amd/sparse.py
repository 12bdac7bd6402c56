"""SparseTensor + coordinate levels: the part of MinkowskiEngine's tensor / coordinate-manager surface that PCGCv2's
encode/decode path touches: `.C .F .tensor_stride .device len()`.

A `CoordMap` is one coordinate level (ME: a coordinate-map key inside the coordinate manager).  It owns the level's
hash table and caches the kernel maps built on it, so all convolutions of a level share one k3 map — ME does the same.
"""
import torch

from . import ops
from ._lib import PcgcError


def require_gpu(device):
    device = torch.device(device)
    if device.type != 'cuda':
        raise PcgcError(f'pcgcv2_amd runs on an MI355X (torch device "cuda" on ROCm); got "{device}". '
                        'There is no CPU execution path.')
    return device


HASH_LEVEL_MAX = 1 << 15     # levels up to this many sites probe the hash for their k3 map; larger ones derive it


class CoordMap:
    """One coordinate level.  `origin` records how the level was produced, which decides how its k3 kernel map is
    built: hash probes only for small / root levels; otherwise a gather through the parent level's map.
         None                                  raw coordinates (root): hash if small, else via its own strided pyramid
         ('children', parent)                  rows 8*i+j of a generative transpose
         ('pruned', cand, mask, prefix)        surviving rows of `cand` (MinkowskiPruning)
         ('selected', cand, bits, wprefix, orig)   the same, written by the one-sweep prune_voxel (ops.topk_select: rank bitmap + orig)
    A children level's coordinates are LAZY (`lazy=(rows, device)`): its convolutions run through the parent level's map and the one-sweep
    pruning derives the survivors' coordinates from the parent's, so the [8 n, 4] tensor only exists if somebody reads `.C`.
    """

    def __init__(self, coords, stride, unique=False, origin=None, lazy=None):
        self._C = coords
        self._n, self.device = (int(lazy[0]), lazy[1]) if coords is None else (coords.shape[0], coords.device)
        self.stride = int(stride)
        self.origin = origin
        self._table = None
        self._k3 = None
        self._down = None
        self._parent_of = None
        self._unique = unique
        self._prepared_up = None
        self._batch_rows = None
        self.descents = None                  # (caller-built levels only: see SparseTensor)

    def __len__(self):
        return self._n

    @property
    def C(self):
        if self._C is None:                                     # a children level nobody has asked the coordinates of so far
            parent = self.origin[1]
            self._C = ops.coords_children(parent.C, parent.stride)
        return self._C

    @property
    def batch_rows(self):
        """Rows per batch item, in item order (ME.utils.sparse_collate puts the item index in column 0; the rows of an item are
        contiguous on every level: canonical orders are first-occurrence orders of an item-contiguous input).  Levels produced by
        up() / pruning inherit theirs; otherwise one device histogram + read-back, cached."""
        if self._batch_rows is None:
            self._batch_rows = ops.batch_counts(self.C) if len(self) else [0]
        return self._batch_rows

    @property
    def table(self):
        if self._table is None:
            self._table = ops.HashTable(self.C, self.stride)
        return self._table

    @property
    def k3(self):
        """[27, N] kernel map of MinkowskiConvolution(kernel_size=3, stride=1) on this level."""
        if self._k3 is None:
            kind = self.origin[0] if self.origin else None
            if len(self) == 0:
                self._k3 = torch.empty((27, 0), dtype=torch.int32, device=self.device)
            elif kind == 'children':
                self._k3 = ops.kmap_k3_children(self.origin[1].k3)
            elif kind == 'pruned':
                _, cand, mask, prefix = self.origin
                orig = ops.compact_index(mask, prefix, len(self))
                if cand._k3 is None and cand.origin is not None and cand.origin[0] == 'children':
                    # candidates = a children level whose own map was never needed (its convs ran through the parent map)
                    self._k3 = ops.kmap_k3_prune_parent(cand.origin[1].k3, mask, prefix, orig)
                else:
                    self._k3 = ops.kmap_k3_prune(cand.k3, mask, prefix, orig)
            elif kind == 'selected':
                _, cand, bits, wprefix, orig = self.origin
                if cand._k3 is None and cand.origin is not None and cand.origin[0] == 'children':
                    self._k3 = ops.kmap_k3_prune_parent_sel(cand.origin[1].k3, bits, wprefix, orig)
                else:
                    self._k3 = ops.kmap_k3_prune_sel(cand.k3, bits, wprefix, orig)
            elif len(self) > HASH_LEVEL_MAX and self.stride <= (1 << 18):
                coarse, down = self.down()
                self._k3 = ops.kmap_k3_from_coarse(self.C, self.stride, self._parent_of, coarse.k3, down)
            else:
                self._k3 = ops.kmap_k3(self.C, self.stride, self.table)
        return self._k3

    def mapless_unit_conv(self):
        """True when a unit-input k3 conv on this level should derive presence from the parent level's map instead of this level's own
        (ops.conv_unit_from_coarse): a raw level that would derive its map from its strided pyramid anyway, and whose map nobody has
        built so far."""
        return (ops.PATH.UNIT_CONV_MAPLESS and self._k3 is None and self.origin is None and len(self) > HASH_LEVEL_MAX and self.stride <= (1 << 18))

    def down(self):
        """-> (coarse CoordMap at 2*stride, [8, N_coarse] kernel map): MinkowskiConvolution(kernel_size=2, stride=2).
        One hash insert + one probe per fine row (the dedup of the quantised coordinates); the map itself is a scatter."""
        if self._down is None:
            coarse_C, self._parent_of, down = ops.down_level(self.C, self.stride)
            self._down = (CoordMap(coarse_C, 2 * self.stride, unique=True), down)
        return self._down

    def build_pyramid(self, levels):
        """The next `levels` strided levels at once (one host synchronisation instead of one per level); fills the same caches
        down() fills and returns the coarsest level.  Levels that are cached already are kept."""
        chain, lvl = [], self
        while lvl._down is not None and len(chain) < levels:
            lvl = lvl._down[0]; chain.append(lvl)
        todo = levels - len(chain)
        if todo > 0:
            for coarse_C, parent_of, down in ops.pyramid(lvl.C, lvl.stride, todo):
                lvl._parent_of = parent_of
                nxt = CoordMap(coarse_C, 2 * lvl.stride, unique=True)
                lvl._down = (nxt, down)
                lvl = nxt
        return lvl

    def up(self):
        """-> children CoordMap at stride/2, rows 8*i+k: MinkowskiGenerativeConvolutionTranspose(k=2, stride=2).
        Not cached on the parent: the child keeps a strong reference to its parent (to derive its kernel map), and a
        back-reference would form a cycle that keeps hundreds of MB of maps alive until Python's cyclic GC runs.
        (`prepare_up()` parks one prebuilt child here; it is handed out — and the reference dropped — by the next up().)"""
        if self._prepared_up is not None:
            child, self._prepared_up = self._prepared_up, None
            return child
        child = CoordMap(None, self.stride // 2, unique=True, origin=('children', self), lazy=(8 * len(self), self.device))
        if self._batch_rows is not None:
            child._batch_rows = [8 * r for r in self._batch_rows]
        return child

    def prepare_up(self):
        """Build the children level and its k3 kernel map ahead of time (the decoder's first stage needs both); used to
        overlap this coordinate-only work with host-side entropy decoding."""
        if (self.origin is None and self._table is None and self._k3 is None and self._prepared_up is None and 0 < len(self) <= HASH_LEVEL_MAX
                and self.stride >= 2 and self._batch_rows is None):
            # a freshly decoded level: everything in one library call (the GPU is waiting for exactly these launches)
            self._table, self._k3, kids, kids_k3 = ops.level_prepare_children(self.C, self.stride)
            child = CoordMap(kids, self.stride // 2, unique=True, origin=('children', self))
            child._k3 = kids_k3
            self._prepared_up = child
            return
        child = self.up()
        child.k3
        self._prepared_up = child

    def drop_caches(self):
        self._table = self._k3 = self._down = self._parent_of = self._prepared_up = None
        self.__dict__.pop('_ingested', None)                   # (coder.Coder._ingest: the sorted copy of an unordered cloud carries caches of its own)


def dedup(coords, feats, stride):
    """ME.SparseTensor construction collapses duplicate coordinates; canonical: keep the first occurrence."""
    from . import conventions
    table = ops.HashTable(coords, stride, keep_last=conventions.get('dedup_keep') == 'last')
    keep = ops.first_occurrence_mask(coords, table)
    prefix, total = ops.mask_scan(keep)
    n = int(total.item())
    if n == coords.shape[0]:
        return coords, feats
    return ops.compact_coords(coords, keep, prefix, n), ops.compact_feats(feats.contiguous(), keep, prefix, n)


class SparseTensor:
    """Drop-in for the ME.SparseTensor uses in coder.py / data_utils.py (ctor: data_utils.py:96,108,116; coder.py:102)."""

    def __init__(self, features, coordinates=None, tensor_stride=1, device=None, coordinate_map=None, assume_unique=False):
        if isinstance(tensor_stride, (list, tuple)):
            tensor_stride = tensor_stride[0]
        self._F_thunk = None
        self._F_rows = None                   # (source tensor, row list): the features are rows of another tensor, not gathered yet
        self.unit_features = False            # True: one channel, every value exactly 1.0 (the occupancy indicator of a coded cloud)
        if coordinate_map is not None:
            self.cmap = coordinate_map
            dev = coordinate_map.device
            if callable(features):                     # deferred features: produced on first access of .F (MinkowskiPruning)
                self._F, self._F_thunk = None, features
                return
            self.F = features if features.device == dev else features.to(dev)
            if self.F.shape[0] != len(coordinate_map):
                raise PcgcError('coordinates / features length mismatch')
        else:
            dev = require_gpu(device if device is not None else coordinates.device)
            coords = coordinates.to(device=dev, dtype=torch.int32).contiguous()
            feats = features.to(device=dev, dtype=torch.float32).contiguous()
            if coords.dim() != 2 or coords.shape[1] != 4:
                raise PcgcError('coordinates must be [N,4] (batch, x, y, z)')
            if coords.shape[0] != feats.shape[0]:
                raise PcgcError('coordinates / features length mismatch')
            descents = ops.check_coords(coords) if coords.shape[0] > 0 else 0      # caller-supplied coordinates: reject what the hash cannot key
            if not assume_unique and coords.shape[0] > 0:
                coords, feats = dedup(coords, feats, int(tensor_stride))
            self.cmap = CoordMap(coords, int(tensor_stride), unique=True)
            self.cmap.descents = descents                  # how unordered the caller's rows are (Coder._ingest sorts an unordered cloud once)
            self.F = feats
            # (decided once, here, where the constructor synchronises anyway: the first layer then needs no feature gathers; the
            #  claim is tied to this very tensor in this very state — see has_unit_features)
            if feats.shape[1] == 1 and feats.shape[0] > 0 and bool((feats == 1).all().item()):
                self.unit_features = True
                self._unit_stamp = (feats.data_ptr(), feats._version)

    @property
    def F(self):
        if self._F is None and self._F_thunk is not None:
            self._F, self._F_thunk, self._F_rows = self._F_thunk(), None, None
        return self._F

    def pending_rows(self):
        """(source, rows) while the features are still rows `rows` of `source` (a level pruned by the one-sweep prune_voxel whose
        features nobody has read): a consumer that can read them in place (the next stage's transposed convolution) does so and the
        compacted tensor is never written.  None once .F has been materialised."""
        return self._F_rows if self._F is None else None

    @F.setter
    def F(self, value):
        self._F, self._F_thunk, self._F_rows = value, None, None
        self.unit_features = False            # (new features: whatever was known about the old ones is void)

    def has_unit_features(self):
        """True iff the features are STILL the all-ones [N, 1] tensor the constructor saw: same storage, no in-place write since
        (`x.F.mul_(2)`, `x.F[...] = v` bump the tensor's version; `x.F = other` clears the flag).  MinkowskiEngine and the reference
        honour the actual feature values; the gather-free first layer (nn.MinkowskiConvolution -> ops.conv_gather_unit) is only
        taken while this holds."""
        f = self._F
        return bool(self.unit_features and f is not None and getattr(self, '_unit_stamp', None) == (f.data_ptr(), f._version))

    @property
    def C(self):
        return self.cmap.C

    @property
    def tensor_stride(self):
        return [self.cmap.stride] * 3

    @property
    def device(self):
        return self.cmap.device

    @property
    def shape(self):
        return self.F.shape

    def __len__(self):
        return len(self.cmap)

    def __repr__(self):
        return f'SparseTensor(N={len(self)}, C={self.F.shape[1]}, stride={self.cmap.stride}, device={self.device})'


def sparse_collate(coords_list, feats_list):
    """ME.utils.sparse_collate (data_utils.py:107,115): prepend the batch index column."""
    cs, fs = [], []
    for b, (c, f) in enumerate(zip(coords_list, feats_list)):
        c = torch.as_tensor(c).int()
        cs.append(torch.cat([torch.full((len(c), 1), b, dtype=torch.int32, device=c.device), c], dim=1))
        fs.append(torch.as_tensor(f))
    return torch.cat(cs, 0), torch.cat(fs, 0)

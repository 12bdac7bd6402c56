"""ONE frozen record of every A/B switch and row-count gate of the kernel dispatch (pcgcv2_amd/dispatch.py).

Rounds 1-5 kept these as a dozen mutable module attributes of `ops` that tests flipped one by one and an autouse fixture had to put back.
They are fields of an immutable `PathConfig` now: `dispatch.select(..., cfg)` is a pure function of it, the process-wide default lives in
ONE place (`ops.PATH`, built from the environment once), and changing it means REPLACING it (`ops.configure(FIELD=value)`, or the
context manager `ops.path(FIELD=value)` that restores the previous record on exit).  For the A/B tools and tests written against the old
spelling, `ops.FIELD` still reads the current record's field and `ops.FIELD = value` is `ops.configure(FIELD=value)` — a replacement of the
record, never a mutation of it; nothing in the product path writes either.

Every family of one operator computes the same canonical fmaf chain: the record affects speed, never a bit of the result."""
import dataclasses
import os


def _env(name, default='1'):
    return os.environ.get(name, default) != '0'


@dataclasses.dataclass(frozen=True)
class PathConfig:
    # first layer on the all-ones occupancy input: no feature gathers (k_conv_unit); and, on a pyramid level, no kernel map of its own
    UNIT_INPUT_CONV: bool = True
    UNIT_CONV_MAPLESS: bool = True
    # InceptionResNet as two fused passes (off: the five-conv composition the tests compare against)
    FUSE_IRN: bool = True
    # C = 64 blocks through the level's own map: LDS-resident table, one wave per 16-row tile (csrc/rows_irn.hip); rows from which it is taken
    ROWS_IRN64: bool = True
    ROWS_IRN64_MIN: int = 1024            # (tools/rows_gate_ab.py: 65 vs 135 us per block at 1.1-18 k rows, 103 vs 198 at 71 k)
    # C = 32 blocks on plain levels, packed-N rows kernels; upper bound = the kernels' 32-bit buffer offsets (n * 32 * 4 B < 0xF0000000)
    ROWS_IRN32: bool = True
    ROWS_IRN32_MIN: int = 1024            # (47 vs 74 us per block at 49 k rows, 119 vs 127 at 256 k against the VALU passes)
    ROWS_IRN32_MAX: int = 0xF0000000 // (32 * 4) - 1
    # C = 32 blocks on LARGE plain levels in quad-block form (csrc/rows_q4.hip).  The packed-N kernels keep the small levels: their 16-row
    # tiles fill the chip where 64-row tiles are a single round of lone waves (tools/rows32_ab.py, us per block, packed-N vs quad-block:
    # 18.7 k rows 30 vs 43, 71 k rows 40 vs 46, 256 k rows 120 vs 87)
    ROWS_Q4: bool = True
    ROWS_Q4_MIN: int = 150_000
    # k3 convs / InceptionResNets on children levels through the PARENT map (csrc/child_kernels.h)
    CHILD_MFMA: bool = True
    # C = 16 blocks and the 16 -> 1 head of children levels in quad-block form (csrc/child_q4.h), from this many parents on: one 128-parent tile
    # per wave on 2 048 wave slots — below ~1 600 tiles a lone wave's tile time is what the launch takes (64 k parents: 78 us against the
    # packed-N pass A's 37; 225 k: 97 against 100; 256 k: 97 against 114; tools/child_q4_variants.py)
    CHILD_Q4: bool = True
    CHILD_Q4_MIN_PARENTS: int = 200_000
    # k3 32 -> 32 on plain levels: LDS-resident table, one wave per 16-row tile
    ROWS_CONV: bool = True
    ROWS_CONV_MIN: int = 1024             # (34 vs 51 us at 1.1-18 k rows, 127 vs 159 at 256 k)
    # k3 64 -> 64 with present-row packing (csrc/conv_packed.hip)
    PACKED_CONV64: bool = True
    PACKED_CONV64_MIN: int = 512          # (39 vs 86 us at 1-4 k rows, 44 vs 113 at 8 k, 75 vs 115 at 33 k, 112 vs 164 at 66 k)
    # k2 s2 down convs: LDS-resident table, one wave per 16 coarse rows
    ROWS_DOWN: bool = True
    ROWS_DOWN_MIN: int = 1024
    # prune_voxel as radix passes + one scan (csrc/select.hip, pcgc_topk_select)
    ONE_SWEEP_PRUNE: bool = True
    # D1 nearest neighbours through 4 x 4 x 4 cells with occupancy masks (off: one probe per lattice offset)
    D1_CELLS: bool = True

    @classmethod
    def from_env(cls):
        """the defaults, with the switches an environment variable PCGC_<FIELD>=0 turns off (A/B runs of whole commands: bench.py, tools/)"""
        off = {f.name: False for f in dataclasses.fields(cls) if f.type in (bool, 'bool') and not _env('PCGC_' + f.name)}
        return cls(**off)

    def replace(self, **changes):
        unknown = set(changes) - FIELDS
        if unknown:
            raise AttributeError(f'PathConfig has no field(s) {sorted(unknown)}')
        return dataclasses.replace(self, **changes)

    def switches(self):
        """the boolean fields (bench.py prints them on its line)"""
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self) if isinstance(getattr(self, f.name), bool)}


FIELDS = frozenset(f.name for f in dataclasses.fields(PathConfig))

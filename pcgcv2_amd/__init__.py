"""pcgcv2_amd — MI355X-native encode/decode path of PCGCv2 (NJUVISION/PCGCv2) behind the reference's coder.py /
pcc_model.py API and bitstream.  Hot ops live in libpcgc_hip.so (hand-written HIP for gfx950, include/pcgc_hip.h)."""
from ._lib import PcgcError, LIB_PATH  # noqa: F401

__all__ = ['PcgcError', 'LIB_PATH']

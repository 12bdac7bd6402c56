"""Lossless coding of the stride-8 coordinates (`_C.bin`).

Reference: gpcc.py:6-41 shells out to MPEG G-PCC `tmc3` (TMC13 v12) through two temporary ASCII PLY files.  The binary
is an external, un-vendored program; if one is installed (env PCGC_TMC3 or ./tmc3 next to this file) the same command
lines are used for interoperability.  Otherwise the native octree codec of libpcgc_hip.so (pcgc_oct_encode /
pcgc_oct_decode, magic "PCGO", NOT G-PCC compatible) is used and no temp files or processes are involved."""
import os
import subprocess
import numpy as np

from . import ops

rootdir = os.path.split(__file__)[0]


def tmc3_path():
    p = os.environ.get('PCGC_TMC3') or os.path.join(rootdir, 'tmc3')
    return p if os.path.isfile(p) and os.access(p, os.X_OK) else None


def gpcc_encode(filedir, bin_dir, show=False):
    """gpcc.py:6-27 (same flags)."""
    exe = tmc3_path()
    if exe is None:
        raise FileNotFoundError('tmc3 binary not found (set PCGC_TMC3); the native codec is used through CoordinateCoder')
    cmd = [exe, '--mode=0', '--positionQuantizationScale=1', '--trisoupNodeSizeLog2=0', '--neighbourAvailBoundaryLog2=8',
           '--intra_pred_max_node_size_log2=6', '--inferredDirectCodingMode=0', '--maxNumQtBtBeforeOt=4',
           '--uncompressedDataPath=' + filedir, '--compressedStreamPath=' + bin_dir]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if show:
        print(r.stdout.decode(errors='replace'))
    if r.returncode != 0:
        raise RuntimeError('tmc3 encode failed: ' + r.stdout.decode(errors='replace')[-500:])


def gpcc_decode(bin_dir, rec_dir, show=False):
    """gpcc.py:29-41."""
    exe = tmc3_path()
    if exe is None:
        raise FileNotFoundError('tmc3 binary not found (set PCGC_TMC3)')
    cmd = [exe, '--mode=1', '--compressedStreamPath=' + bin_dir, '--reconstructedDataPath=' + rec_dir, '--outputBinaryPly=0']
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if show:
        print(r.stdout.decode(errors='replace'))
    if r.returncode != 0:
        raise RuntimeError('tmc3 decode failed: ' + r.stdout.decode(errors='replace')[-500:])


def native_encode(coords, bin_dir):
    with open(bin_dir, 'wb') as f:
        f.write(ops.oct_encode(np.asarray(coords, dtype=np.int32)))


def native_decode(bin_dir):
    with open(bin_dir, 'rb') as f:
        return ops.oct_decode(f.read()).astype('int')


def is_native_stream(bin_dir):
    with open(bin_dir, 'rb') as f:
        return f.read(4) == b'PCGO'

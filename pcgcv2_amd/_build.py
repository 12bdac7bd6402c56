"""Builds pcgcv2_amd/libpcgc_hip.so (gfx950 only) with hipcc.  In-tree so the .so travels with the repo snapshot."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
# PCGC_BUILD_VARIANT=<name> (experiments only): a second library libpcgc_hip_<name>.so from objects in csrc/_obj_<name>/, built with
# PCGC_EXTRA_HIPCC_FLAGS — e.g. the per-phase cycle counters of the children-level kernels (-DPCGC_CHILD_TIMING) next to the product
# build; load it with PCGC_LIB=<path> (pcgcv2_amd/_lib.py).  Objects whose source does not include child_kernels.h are copied over.
VARIANT = os.environ.get('PCGC_BUILD_VARIANT', '')
OBJ = os.path.join(CSRC, '_obj' + ('_' + VARIANT if VARIANT else ''))
LIB = os.path.join(HERE, 'libpcgc_hip' + ('_' + VARIANT if VARIANT else '') + '.so')
SOURCES = ['coords.hip', 'select.hip', 'conv.hip', 'child_conv.hip', 'child_conv32.hip', 'child_cls_w.hip', 'child_irn.hip', 'child_irn_a16.hip', 'child_irn_b16.hip',
           'child_irn_a32.hip', 'child_irn_b32.hip', 'child_q4.hip', 'rows_irn.hip', 'rows_q4.hip', 'conv_packed.hip', 'entropy.hip', 'hostcodec.cpp', 'ply.cpp']
HEADERS = [os.path.join(CSRC, 'pcgc_common.h'), os.path.join(CSRC, 'mfma_util.h'), os.path.join(CSRC, 'child_kernels.h'), os.path.join(CSRC, 'child_q4.h'), os.path.join(CSRC, 'q4x.h'), os.path.join(CSRC, 'q4x_sched.h'), os.path.join(CSRC, 'rows_q4_policy.h'), os.path.join(HERE, '..', 'include', 'pcgc_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-Wall', '-Wno-unused-result'] + \
        os.environ.get('PCGC_EXTRA_HIPCC_FLAGS', '').split()          # (experiments only, e.g. -DPCGC_EXP_...)


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    if VARIANT:
        import shutil
        base = os.path.join(CSRC, '_obj')
        for src in SOURCES:
            o = src.rsplit('.', 1)[0] + '.o'
            if 'child_kernels.h' not in open(os.path.join(CSRC, src)).read() and os.path.exists(os.path.join(base, o)) \
                    and not os.path.exists(os.path.join(OBJ, o)):
                shutil.copy2(os.path.join(base, o), os.path.join(OBJ, o))
    hipcc = _hipcc()
    jobs = []
    only = [t for t in os.environ.get('PCGC_BUILD_ONLY', '').split(',') if t]      # (kernel experiments: rebuild just these units,
    for src in SOURCES:                                                             #  e.g. PCGC_BUILD_ONLY=child_irn_a16 — seconds instead of minutes)
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.rsplit('.', 1)[0] + '.o')
        if only and os.path.exists(o):
            if src.rsplit('.', 1)[0] in only:
                jobs.append((o, [hipcc] + FLAGS + (['-x', 'hip'] if src.endswith('.hip') else []) + ['-c', s, '-o', o]))
            continue
        if force or _stale(o, [s] + HEADERS):
            lang = ['-x', 'hip'] if src.endswith('.hip') else []
            jobs.append((o, [hipcc] + FLAGS + lang + ['-c', s, '-o', o]))

    def run(job):
        o, cmd = job
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + ' '.join(cmd) + '\n' + r.stderr[-4000:])
        return o

    with ThreadPoolExecutor(max_workers=max(2, min(8, os.cpu_count() or 4))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.rsplit('.', 1)[0] + '.o') for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-lz']
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr[-4000:])
    if not VARIANT:
        build_reftable(force=force)
    return LIB


REFTABLE_LIB = os.path.join(HERE, 'libpcgc_reftable.so')


def build_reftable(force=False):
    """libpcgc_reftable.so: the reference-arithmetic CDF table as ATen operators issued from C++ (csrc/reftable.cpp).  Host code,
    plain g++ against the torch headers / libtorch_cpu of this environment."""
    src = os.path.join(CSRC, 'reftable.cpp')
    if not (force or _stale(REFTABLE_LIB, [src])):
        return REFTABLE_LIB
    import torch
    tdir = os.path.dirname(torch.__file__)
    cmd = ['g++', '-O2', '-std=c++17', '-fPIC', '-shared', f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}',
           '-I' + os.path.join(tdir, 'include'), '-I' + os.path.join(tdir, 'include', 'torch', 'csrc', 'api', 'include'), src,
           '-o', REFTABLE_LIB, '-L' + os.path.join(tdir, 'lib'), '-ltorch_cpu', '-lc10', '-Wl,-rpath,' + os.path.join(tdir, 'lib')]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('g++ failed:\n' + ' '.join(cmd) + '\n' + r.stderr[-4000:])
    return REFTABLE_LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))

"""pcgcv2_amd — MI355X-native encode/decode path of PCGCv2 (NJUVISION/PCGCv2) behind the reference's coder.py /
pcc_model.py API and bitstream.  Hot ops live in libpcgc_hip.so (hand-written HIP for gfx950, include/pcgc_hip.h)."""
from ._lib import PcgcError, LIB_PATH  # noqa: F401

__all__ = ['PcgcError', 'LIB_PATH']


def effective_cpus():
    """CPUs this process may actually use: min(affinity, cgroup cpu.max quota).  Containers often advertise all host
    cores (os.cpu_count()) while a CFS quota allows far fewer; sizing thread pools from cpu_count then gets the process
    throttled for the rest of every 100 ms period."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def configure_host_threads(max_threads=4, local_world=None):
    """The host side of the codec is a single-threaded launcher + sequential entropy coder (+ a few short-lived helper threads for the
    indexed entropy decoders); keep torch's CPU thread pool small so its workers do not spin away the container's CPU quota, and size
    the decoder pools from this process's SHARE of the CPUs: `local_world` = processes on this node that share them (one rank per GPU;
    default: LOCAL_WORLD_SIZE or 1)."""
    import os
    import torch
    if local_world is None:
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1)
    cpus = max(1, effective_cpus() // max(1, int(local_world)))
    torch.set_num_threads(max(1, min(max_threads, cpus)))
    from . import ops
    ops.set_rc_threads(max(1, min(8, cpus - 2)))                 # segments of an indexed `_F.bin` / groups of `_C.bin` decoded side by side

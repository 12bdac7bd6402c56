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


def _parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def numa_cpus_of_gpu(pci_bus_id, sysfs='/sys'):
    """CPUs of the NUMA node a GPU hangs off, from sysfs (`<sysfs>/bus/pci/devices/<domain:bus:dev.fn>/numa_node` ->
    `<sysfs>/devices/system/node/node<N>/cpulist`); None if the platform does not say (single node, or numa_node = -1)."""
    import os
    try:
        node = int(open(os.path.join(sysfs, 'bus/pci/devices', pci_bus_id.lower(), 'numa_node')).read())
        if node < 0:
            return None
        return _parse_cpulist(open(os.path.join(sysfs, f'devices/system/node/node{node}/cpulist')).read()) or None
    except (OSError, ValueError):
        return None


def pin_to_gpu_numa_node(device_index):
    """Restrict this process (one rank per GPU) to the CPUs of its GPU's NUMA node: the host stages of the codec — range coder,
    octree coder, table evaluation, pinned staging copies — then run next to the memory and the PCIe root port they use, and
    the eight ranks of a node do not migrate across sockets.  -> the CPU list applied, or None if nothing was changed (unknown
    topology, or the intersection with the current affinity mask is empty)."""
    import os
    import torch
    if not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        p = torch.cuda.get_device_properties(device_index)
        bus = f'{getattr(p, "pci_domain_id", 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0'
    except (AttributeError, RuntimeError, AssertionError):
        return None
    cpus = numa_cpus_of_gpu(bus)
    if not cpus:
        return None
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
    if not allowed:
        return None
    os.sched_setaffinity(0, allowed)
    return allowed


def configure_host_threads(max_threads=1, local_world=None, frames_in_flight=1, pin_device=None):
    """The host side of the codec is a single-threaded launcher + sequential entropy coder (+ a few short-lived helper threads for the
    indexed entropy decoders); keep torch's CPU thread pool small so its workers do not spin away the container's CPU quota, and size
    the decoder pools from this process's SHARE of the CPUs: `local_world` = processes on this node that share them (one rank per GPU;
    default: LOCAL_WORLD_SIZE or 1), `frames_in_flight` = frames this process codes concurrently (shard.code_units(in_flight=F)):
    the budget is   frames x (1 launcher + 1 coordinate helper + range-decoder helpers + ATen threads)  <=  this process's CPUs.
    `pin_device` (GPU index): also pin the process to that GPU's NUMA-local CPUs (multi-rank nodes).
    `max_threads`: ATen's intra-op threads.  The only ATen CPU work on the path is the CDF table — ~60 operators on [8, 3, ~20] tensors, far
    below any parallel grain size; more threads only add wake-ups (measured: 0.42 ms with one thread, 0.59 with four, cold) — so 1.
    With a budget of one or two range-decoder threads the library switches to its lane-parallel decoder by itself (pcgc_set_rc_lanes).
    -> dict of what was applied."""
    import os
    import torch
    if local_world is None:
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1)
    pinned = pin_to_gpu_numa_node(pin_device) if pin_device is not None else None
    cpus = max(1, effective_cpus() // max(1, int(local_world)))
    per_frame = max(1, cpus // max(1, int(frames_in_flight)))
    aten = max(1, min(max_threads, per_frame - 2)) if frames_in_flight > 1 else max(1, min(max_threads, cpus))
    rc = max(1, min(8, per_frame - 2))
    torch.set_num_threads(aten)
    from . import ops
    ops.set_rc_threads(rc)                                       # segments of an indexed `_F.bin` / groups of `_C.bin` decoded side by side
    return {'cpus': cpus, 'frames_in_flight': int(frames_in_flight), 'aten_threads': aten, 'rc_threads': rc, 'pinned_cpus': pinned}

#!/usr/bin/env python3
"""Aggregate rocprofv3 counter_collection CSVs (FETCH_SIZE / WRITE_SIZE passes) into per-(kernel, grid) HBM bytes per launch.
rocprofv3 reports kilobytes (x1024).  On gfx950 FETCH_SIZE tallies 64 B per memory-side request whatever its size (calibrated on known
byte counts, profiles/r02_fetch_calibration.txt): wide streaming reads and gathers of 128- / 256-byte row segments issue 128-byte
requests and read back HALF their bytes (correction x2, as MI355X_MICROARCH.md says); gathers of 64-byte rows read back exactly their
bytes (x1); 32-byte rows read back 2x their bytes, which is true traffic (64-byte sectors).  So kernels whose traffic is dominated by
gathers of rows of at most 64 bytes (the C = 16 level) take x1, everything else x2; mixed kernels lie in between."""
import csv, glob, hashlib, json, os, re, sys, collections
src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        m = re.search(r'(k_[a-z0-9_]+<[^>]*>|k_[a-z0-9_]+)', name.replace('(anonymous namespace)::', ''))
        if not m:
            continue
        acc[(m.group(1), int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
out = []
for (k, grid), d in acc.items():
    if 'FETCH_SIZE' not in d or 'WRITE_SIZE' not in d:
        continue
    fetch = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']) * 1024
    write = sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE']) * 1024
    narrow = bool(re.match(r'k_(irn_[ab]<16|child_irn_[ab]<16|child_q4<|child_conv<1, 1|child_cls<1,|conv_gather_\w+<16,)', k))
    f = 1.0 if narrow else 2.0
    out.append({'kernel': k, 'grid_size': grid, 'grid_rows': grid, 'launches_sampled': len(d['FETCH_SIZE']),
                'fetch_bytes_raw': round(fetch), 'fetch_correction': f, 'fetch_bytes_corrected': round(f * fetch), 'write_bytes': round(write),
                'hbm_bytes_per_launch': round(f * fetch + write),
                'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py; FETCH_SIZE x%g (calibrated: 64 B tallied per '
                          'request; %s)' % (f, '64-byte row gathers read back their bytes' if narrow else '128-byte requests read back half')})
out.sort(key=lambda e: -e['hbm_bytes_per_launch'])


def kernel_sources_sha16(root):
    """hash of every kernel source of the library: bench.py replays this file only while it matches the sources it runs"""
    h = hashlib.sha256()
    d = os.path.join(root, 'pcgcv2_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):            # device sources (the host codec, ply and table .cpp files launch nothing)
            h.update(f.encode()); h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump({'kernel_sources_sha16': kernel_sources_sha16(root), 'collected_by': 'tools/pmc_traffic.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py)',
           'kernels': out}, open(dst, 'w'), indent=1)
for e in out[:12]:
    print(e['kernel'], e['grid_size'], 'fetch(raw) %.1f MB  write %.1f MB' % (e['fetch_bytes_raw'] / 1e6, e['write_bytes'] / 1e6))

#!/bin/bash
# FETCH_SIZE calibration on known byte counts (tools/ubench/fetch_calib.hip) -> profiles/fetch_calibration.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -x $R/tools/ubench/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/ubench/fetch_calib $R/tools/ubench/fetch_calib.hip > /tmp/fc_build.log 2>&1
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
rm -rf /tmp/fc; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc -- $R/tools/ubench/fetch_calib > /tmp/fc.log 2>&1
python - <<PY > $R/gpurun_out/fetch_calibration.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/fc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE': acc[r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
known = 2 << 30
print('# rocprofv3 --pmc FETCH_SIZE on tools/ubench/fetch_calib (every kernel reads exactly 2 GiB = %d bytes, once, from HBM)' % known)
print('# kernel                                   FETCH_SIZE(raw, bytes)   raw/known   correction factor to apply')
for k, v in sorted(acc.items()):
    raw = sum(v) / len(v) * 1024
    print(f'{k:42s} {raw:18.0f}   {raw / known:8.4f}   x{known / raw:6.3f}')
PY
cat $R/gpurun_out/fetch_calibration.txt

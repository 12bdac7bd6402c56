#!/bin/bash
# Which kernels of libpcgc_hip.so are ever launched?  (a) by the four bench configurations with default switches, (b) by the whole GPU test suite
# (which also walks the A/B switches).  Writes gpurun_out/zoo/{bench,tests}_kernels.txt: "<calls> <kernel name>" per distinct kernel.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/zoo
mkdir -p $O
names() {   # directory of a rocprofv3 run -> calls + name per kernel, from the kernel-trace CSVs (the stats file truncates nothing either, but may be absent)
    python3 - "$1" <<'PY'
import csv, glob, sys, collections
c = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    with open(f, newline='') as fh:
        for row in csv.DictReader(fh):
            c[row['Kernel_Name']] += 1
for k, v in sorted(c.items()):
    print(v, k)
PY
}
cd $R
for cfg in frame batch4 sweep blocks; do
    rocprofv3 --kernel-trace -d /tmp/zoo_$cfg -o t --output-format csv -- python bench.py --config $cfg --steps 2 --warmup 1 --no-extra --no-cpu-baseline > /tmp/zoo_$cfg.log 2>&1
    names /tmp/zoo_$cfg > $O/bench_${cfg}_kernels.txt
    rm -rf /tmp/zoo_$cfg
done
rocprofv3 --kernel-trace -d /tmp/zoo_noisy -o t --output-format csv -- python bench.py --config frame --workload noisy10 --steps 2 --warmup 1 --no-extra --no-cpu-baseline > /tmp/zoo_noisy.log 2>&1
names /tmp/zoo_noisy > $O/bench_noisy10_kernels.txt
rm -rf /tmp/zoo_noisy
rocprofv3 --kernel-trace -d /tmp/zoo_tests -o t --output-format csv -- python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1
names /tmp/zoo_tests > $O/tests_kernels.txt
tail -3 $O/tests.log
wc -l $O/*_kernels.txt

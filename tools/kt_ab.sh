# kernel-trace A/B of the bench frame: usage  tools/kt_ab.sh <tag> [ENV=VAL ...]   -> gpurun_out/kt_ab/<tag>.txt (per-kernel summary of the timed steps)
R=$GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p $R/gpurun_out/kt_ab
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$tag
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python $R/bench.py --config frame --steps 10 --warmup 2 --no-cpu-baseline --no-events --serving-frames 0 --no-extra > $R/gpurun_out/kt_ab/$tag.log 2>&1
cd $R
python tools/trace_window_summary.py /tmp/kt_$tag/*/*kernel_trace.csv 10 1 > gpurun_out/kt_ab/$tag.txt 2>&1
head -3 gpurun_out/kt_ab/$tag.txt

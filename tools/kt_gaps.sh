R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6d
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6d/kt -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-events --serving-frames 0 --no-extra > $R/gpurun_out/r6d/kt.log 2>&1
cd $R
python tools/trace_window_summary.py gpurun_out/r6d/kt/*/*kernel_trace.csv 10 1 > gpurun_out/r6d/kernel_trace.txt 2>&1
python tools/trace_gaps.py gpurun_out/r6d/kt/*/*kernel_trace.csv > gpurun_out/r6d/gaps.txt 2>&1
find gpurun_out/r6d/kt -name '*kernel_trace.csv' -delete
head -30 gpurun_out/r6d/gaps.txt

set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
timeout 600 bash $R/tools/pmc_traffic.sh > $R/gpurun_out/final/pmc.log 2>&1
cp $R/profiles/pmc_traffic.json $R/gpurun_out/final/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-events --serving-frames 0 > $R/gpurun_out/final/kt.log 2>&1
cd $R
python tools/rocprof_summary.py gpurun_out/final/kt/*/*kernel_stats.csv at::native > gpurun_out/final/kernel_trace.txt 2>&1 || true
find gpurun_out/final/kt -name '*kernel_trace.csv' -delete
python bench.py --steps 20 --warmup 5 --detail gpurun_out/final/detail.json > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench.err
tail -c 1500 gpurun_out/final/bench_line.json

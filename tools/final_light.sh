set -x
# The part of tools/final_collect.sh that depends on the kernel SOURCES' final state: PMC traffic (stamped with the source hash), the frame's kernel
# trace + idle gaps, and the bench lines of every configuration.  -> gpurun_out/final2/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final2; mkdir -p $O
timeout 600 bash $R/tools/pmc_traffic.sh > $O/pmc.log 2>&1
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_frame -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-events --serving-frames 0 --no-extra > $O/kt_frame.log 2>&1
cd $R
python tools/trace_window_summary.py $O/kt_frame/*/*kernel_trace.csv 10 1 > $O/kernel_trace_frame.txt 2>&1 || true
python tools/rocprof_summary.py $O/kt_frame/*/*kernel_stats.csv > $O/kernel_stats_frame.txt 2>&1 || true
python tools/trace_gaps.py $O/kt_frame/*/*kernel_trace.csv > $O/gpu_idle_gaps.txt 2>&1
find $O/kt_frame -name '*kernel_trace.csv' -delete
python bench.py --steps 20 --warmup 5 --detail $O/detail.json > $O/bench_frame.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_frame_again.json 2>> $O/bench.err
python bench.py --config batch4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_batch4.json 2>> $O/bench.err
python bench.py --config sweep --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_sweep.json 2>> $O/bench.err
python bench.py --config blocks --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_blocks.json 2>> $O/bench.err
python bench.py --workload noisy10 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_noisy10.json 2>> $O/bench.err
tail -c 400 $O/bench_frame.json

set -x
# Round-end collection on the GPU box: PMC traffic (FETCH_SIZE / WRITE_SIZE passes), rocprofv3 kernel traces of the bench command (frame, sweep, blocks,
# noisy10), SQ / TCC counters of the children-level and rows kernels, FETCH_SIZE calibration, and the bench lines of every config.  -> gpurun_out/final/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
timeout 600 bash $R/tools/pmc_traffic.sh > $R/gpurun_out/final/pmc.log 2>&1
cp $R/profiles/pmc_traffic.json $R/gpurun_out/final/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
for cfg in frame:10:1 sweep:2:7 blocks:3:8; do
  c=${cfg%%:*}; rest=${cfg#*:}; st=${rest%%:*}; per=${rest##*:}
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/kt_$c -- python $R/bench.py --config $c --steps $st --warmup 2 --no-cpu-baseline --no-events --serving-frames 0 --no-extra > $R/gpurun_out/final/kt_$c.log 2>&1
  cd $R
  python tools/trace_window_summary.py gpurun_out/final/kt_$c/*/*kernel_trace.csv $st $per > gpurun_out/final/kernel_trace_$c.txt 2>&1 || true
  python tools/rocprof_summary.py gpurun_out/final/kt_$c/*/*kernel_stats.csv > gpurun_out/final/kernel_stats_$c.txt 2>&1 || true
  [ $c = frame ] && python tools/trace_gaps.py gpurun_out/final/kt_$c/*/*kernel_trace.csv > gpurun_out/final/gpu_idle_gaps.txt 2>&1
  find gpurun_out/final/kt_$c -name '*kernel_trace.csv' -delete
done
bash tools/fetch_calib.sh > /dev/null 2>&1; cp gpurun_out/fetch_calibration.txt gpurun_out/final/fetch_calibration.txt 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/kt_noisy10 -- python $R/bench.py --workload noisy10 --steps 10 --warmup 2 --no-cpu-baseline --no-events --serving-frames 0 --no-extra > $R/gpurun_out/final/kt_noisy10.log 2>&1
cd $R
python tools/trace_window_summary.py gpurun_out/final/kt_noisy10/*/*kernel_trace.csv 10 1 > gpurun_out/final/kernel_trace_noisy10.txt 2>&1 || true
find gpurun_out/final/kt_noisy10 -name '*kernel_trace.csv' -delete
python bench.py --workload noisy10 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --detail gpurun_out/final/detail_noisy10.json > gpurun_out/final/bench_noisy10.json 2>> gpurun_out/final/bench.err
bash tools/q4_pmc.sh shell10 > /dev/null 2>&1; cp gpurun_out/q4_pmc/summary.txt gpurun_out/final/child_q4_pmc.txt
for cfg in 16:irn 16:conv 32:irn; do
  bash tools/child_pmc.sh ${cfg%%:*} 0 0 ${cfg##*:} > /dev/null 2>&1; cp gpurun_out/child_pmc/summary.txt gpurun_out/final/child_pmc_${cfg%%:*}_${cfg##*:}.txt
done
python bench.py --steps 20 --warmup 5 --detail gpurun_out/final/detail.json > gpurun_out/final/bench_frame.json 2> gpurun_out/final/bench.err
python bench.py --config batch4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/bench_batch4.json 2>> gpurun_out/final/bench.err
python bench.py --config sweep --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_sweep.json 2>> gpurun_out/final/bench.err
python bench.py --config blocks --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_blocks.json 2>> gpurun_out/final/bench.err
tail -c 1200 gpurun_out/final/bench_frame.json

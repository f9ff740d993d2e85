mkdir -p gpurun_out/r2
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs3 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-step-sweep > $R/gpurun_out/r2/prof_s3.log 2>&1
cd $R; f=$(find /tmp/profs3 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2/kernel_stats_bench_default_b384_s3.csv; head -6 gpurun_out/r2/kernel_stats_bench_default_b384_s3.csv | cut -c1-130; tail -1 gpurun_out/r2/prof_s3.log | cut -c1-200

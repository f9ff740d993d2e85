mkdir -p gpurun_out/r2
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r2/t_call5.log 2>&1; tail -6 gpurun_out/r2/t_call5.log
timeout 120 python profiles/orpheus_bench.py > gpurun_out/r2/orpheus_defaults.log 2>&1; tail -2 gpurun_out/r2/orpheus_defaults.log
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof384 -- python $R/bench.py --batch 384 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/r2/prof384.log 2>&1
cd $R; f=$(find /tmp/prof384 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2/kernel_stats_b384_s1.csv; head -30 gpurun_out/r2/kernel_stats_b384_s1.csv | cut -c1-200

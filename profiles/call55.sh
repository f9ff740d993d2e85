mkdir -p gpurun_out/r2
timeout 900 python bench.py > gpurun_out/r2/bench_final.json 2> gpurun_out/r2/bench_final.log; echo rc=$?
for w in dia orpheus kokoro; do timeout 600 python bench.py --workload $w --steps 2 --warmup 1 > gpurun_out/r2/bench_$w.json 2> gpurun_out/r2/bench_$w.log; echo "$w rc=$?"; done
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proffinal -- python $R/bench.py --batch 384 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-step-sweep > $R/gpurun_out/r2/prof_final.log 2>&1
cd $R; f=$(find /tmp/proffinal -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2/kernel_stats_bench_b384_s1_final.csv; head -4 gpurun_out/r2/kernel_stats_bench_b384_s1_final.csv | cut -c1-120

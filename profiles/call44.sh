mkdir -p gpurun_out/r2
timeout 900 python bench.py --workload dia --steps 2 --warmup 1 > gpurun_out/r2/bench_dia.json 2> gpurun_out/r2/bench_dia.log; echo rc=$?; tail -3 gpurun_out/r2/bench_dia.log; cat gpurun_out/r2/bench_dia.json | cut -c1-1500

mkdir -p gpurun_out/r2
timeout 500 python profiles/perf_battery_run.py > gpurun_out/r2/perf_battery.log 2>&1; tail -12 gpurun_out/r2/perf_battery.log

mkdir -p gpurun_out/r4
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
timeout 900 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "passed|failed|^E |^FAILED|rror|first divergence|worst|identical to" | tail -14 > gpurun_out/r4/gpu_tests_call2.txt
cat gpurun_out/r4/gpu_tests_call2.txt

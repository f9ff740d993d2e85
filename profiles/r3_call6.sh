mkdir -p gpurun_out/r3
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -12 > gpurun_out/r3/gpu_tests_b3_default.txt
cat gpurun_out/r3/gpu_tests_b3_default.txt
timeout 900 python bench.py > gpurun_out/r3/bench_default_b3.json 2> gpurun_out/r3/bench_default_b3.log
tail -c 3000 gpurun_out/r3/bench_default_b3.json

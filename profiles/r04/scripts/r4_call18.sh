mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -6 > gpurun_out/r4/gpu_tests_final2.txt; cat gpurun_out/r4/gpu_tests_final2.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

# last call of the round: the full GPU suite on the final library (attn_short with the prompt length as a compile-time bound)
mkdir -p gpurun_out/r3
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
ulimit -c 0
timeout 130 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 > gpurun_out/r3/gpu_tests_call38.txt
cat gpurun_out/r3/gpu_tests_call38.txt

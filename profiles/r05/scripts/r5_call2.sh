# round 5, call 2: the one-sequence chains with the staging inputs requested before the weights everywhere (gemm16_kernel, gemv_stream_kernel, the Q4_0 kernels):
# whole GPU suite, Parler batch-1 chain with stamps, Dia step, Orpheus step
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x -rP -k "not history" 2>&1 | grep -E "passed|failed|^E |^FAILED|rror|streams identical" | tail -12 | tee $O/gpu_tests_call2.txt
timeout 600 python -m pytest tests/test_gpu_parler.py -m gpu -q -rP -k "history" 2>&1 | grep -E "passed|failed|^E |^FAILED|rror|relative logit" | tail -12 | tee $O/parity_attn_rows_history.txt
B1_ONLY_DEFAULT=1 timeout 600 python profiles/b1_chain.py 2>&1 | tee $O/b1_chain_call2.txt | tail -14
timeout 300 python profiles/dia_bench.py 2>&1 | grep -E "ms" | tee $O/dia_bench_call2.txt
ORPHEUS_BENCH_GREEDY_ONLY=1 timeout 200 python profiles/orpheus_bench.py 2>&1 | grep -E "ms/step" | tee $O/orpheus_bench_call2.txt

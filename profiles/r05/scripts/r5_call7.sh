# round 5, call 7: attn_wave_kernel in the Parler one-sequence chain (key slices at static addresses, one round trip): Parler tests, batch-1 chain with stamps
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parler.py tests/test_gpu_runner.py -q -x -k "not history and not many_rows and not tiled" 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 | tee $O/parler_tests_call7.txt
B1_ONLY_DEFAULT=1 timeout 600 python profiles/b1_chain.py 2>&1 | tee $O/b1_chain_call7.txt | tail -14

# round 5, call 8: conv1d_cout1_kernel (the codec's final conv) with every load of a chunk in flight and the next chunk requested under the dot products:
# codec tests, one 64 x 248-frame pass by kernel class
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dac.py tests/test_gpu_snac.py tests/test_gpu_upstream.py -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 | tee $O/dac_tests_call8.txt
timeout 600 python profiles/dac_bench.py 248 3 --batch=64 --prof 2>&1 | tail -30 | tee $O/dac_bench_call8.txt

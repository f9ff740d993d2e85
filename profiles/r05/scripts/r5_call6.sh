# round 5, call 6: nucleus sampling (top_p < 1) of the Orpheus step on the device: tests; ms per step with top_p 0.9
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_orpheus.py tests/test_gpu_runner.py -q -k "sampler_over_the_full or sampled_generation" 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -12 | tee $O/orpheus_tests_call6.txt
ORPHEUS_BENCH_TOP_P=0.9 timeout 300 python profiles/orpheus_bench.py 2>&1 | grep -E "ms/step|sampl" | tee $O/orpheus_bench_call6.txt

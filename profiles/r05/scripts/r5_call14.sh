# round 5, call 14: engine_bench variant E: the launch chain with every kernel requesting the next launch's weights into its XCD's L2
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/engine_bench $R/profiles/engine_bench.hip 2>&1 | grep -E "error" | head
ENGINE_BENCH_ONLY_E=1 timeout 120 /tmp/engine_bench 28 2>&1 | tee $O/engine_bench_call14.txt
ENGINE_BENCH_ONLY_E=1 timeout 120 /tmp/engine_bench 28 2>&1 | grep "^E" | tee -a $O/engine_bench_call14.txt

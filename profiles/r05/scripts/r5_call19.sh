# round 5, call 19: does the layout of Dia's cross K / V cache bound its attention launch?  profiles/stride_read_bench.hip
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd /tmp && hipcc --offload-arch=gfx950 -O3 -o /tmp/stride_read_bench $R/profiles/stride_read_bench.hip 2>&1 | grep -E "error" | head
timeout 120 /tmp/stride_read_bench 2>&1 | tee $O/stride_read_bench_call19.txt

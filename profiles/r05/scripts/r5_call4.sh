# round 5, call 4: the persistent-step experiment (profiles/engine_bench.hip): launch chain vs one persistent launch with the next phase's weights
# in flight across the boundary, at the Orpheus-3B Q4_0 layer shapes; the boundary alone
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/engine_bench profiles/engine_bench.hip 2>&1 | grep -E "error" | head
for L in 28; do timeout 120 /tmp/engine_bench $L; done 2>&1 | tee $O/engine_bench_call4.txt

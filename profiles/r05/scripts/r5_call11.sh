# round 5, call 11: the tiled-GEMM sweep at 1024 rows again, without the configurations that do not fit the LDS (round 4's BEST lines were launch failures)
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/gemm_bench profiles/gemm_bench.hip 2>&1 | grep -E "error" | head -3
timeout 300 /tmp/gemm_bench 1024 > $O/gemm_tile_sweep_r1024_call11.log 2>&1
grep BEST $O/gemm_tile_sweep_r1024_call11.log
for sh in qkv proj fc1 fc2; do echo "== $sh: five fastest"; grep "^$sh *R=1024" $O/gemm_tile_sweep_r1024_call11.log | grep -v failed | sort -k7 -g | head -5; done

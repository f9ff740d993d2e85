mkdir -p gpurun_out/r4
GEMM_BENCH_NBUF=1 timeout 200 ./profiles/gemm_bench 1024 > gpurun_out/r4/gemm_tile_sweep_r1024_hot.log 2>&1
for s in qkv proj fc1 fc2; do grep "^$s " gpurun_out/r4/gemm_tile_sweep_r1024_hot.log | awk '{for(i=1;i<=NF;i++) if($i=="us" && $(i-1)>2.0) print $(i-1), $0}' | sort -n | head -2 | cut -d' ' -f2-; done

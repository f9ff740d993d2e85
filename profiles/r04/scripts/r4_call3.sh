mkdir -p gpurun_out/r4
{ echo "== overlap_bench"; timeout 120 ./profiles/overlap_bench; echo "== overlap_bench (second run)"; timeout 120 ./profiles/overlap_bench; } > gpurun_out/r4/overlap_bench.txt 2>&1
cat gpurun_out/r4/overlap_bench.txt
timeout 300 ./profiles/gemm_bench 1024 > gpurun_out/r4/gemm_tile_sweep_r1024.log 2>&1
grep BEST gpurun_out/r4/gemm_tile_sweep_r1024.log

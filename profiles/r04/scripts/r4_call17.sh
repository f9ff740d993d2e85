mkdir -p gpurun_out/r4
timeout 300 ./profiles/gemm_bench 1024 > gpurun_out/r4/gemm_tile_sweep_r1024_wave64.log 2>&1
for s in qkv proj fc1 fc2 heads; do grep "^$s " gpurun_out/r4/gemm_tile_sweep_r1024_wave64.log | awk '{for(i=1;i<=NF;i++) if($i=="us" && $(i-1)>2.0) print $(i-1), $0}' | sort -n | head -3 | cut -d' ' -f2-; done
grep -E "128x128/2x2|256x128|128x256|128x64/2x1|64x128/1x2" gpurun_out/r4/gemm_tile_sweep_r1024_wave64.log | grep -E "^(fc1|qkv|fc2)" | awk '{for(i=1;i<=NF;i++) if($i=="us" && $(i-1)>2.0) print $0}' | head -30

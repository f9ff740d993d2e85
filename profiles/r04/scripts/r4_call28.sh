# round 4, call 28 (second run; the first ran tests/test_gpu_orpheus.py: 18 passed): ms per Orpheus-3B Q4_0 step by who merges the split attention's
# key slices — llama_merge (called attn_fold when this ran) = 1 (last workgroup inside the split kernel) / 0 (combine launch, the default again) / 2 (the o projection's staging)
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out/r4
for f in 1 0 2; do echo "llama_merge=$f"; ORPHEUS_BENCH_GREEDY_ONLY=1 ORPHEUS_TUNE=llama_merge=$f timeout 120 python profiles/orpheus_bench.py 2>&1 | grep -E "ms/step"; done | tee gpurun_out/r4/orpheus_attn_fold_call28.txt

# round 4, call 28: the split attention's in-kernel merge (last workgroup of a (row, head)) in the Orpheus step: tests, then ms per step for
# attn_fold = 0 (combine launch) / 1 (in-kernel, default) / 2 (merged by the o projection's staging)
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
mkdir -p gpurun_out/r4
timeout 300 python -m pytest tests/test_gpu_orpheus.py -q -x --tb=short 2>&1 | tail -15
for f in 1 0 2; do echo "attn_fold=$f"; ORPHEUS_TUNE=attn_fold=$f timeout 120 python profiles/orpheus_bench.py 2>&1 | grep -E "ms per decode step"; done | tee gpurun_out/r4/orpheus_attn_fold_call28.txt

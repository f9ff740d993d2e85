# round 5, call 22 (second run): the decode attention of the Orpheus step at short (96..480) and long (1120..1568) histories, same box:
# attn_gqa_split_kernel (attn_wave=0) / attn_gqa_wave_kernel as of call 5 / with rolling slots beyond its first four passes (attn_roll=1)
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_orpheus.py -q -x 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 | tee $O/orpheus_tests_call22.txt
for rep in 1 2; do for t in "attn_wave=0" "attn_roll=0" "attn_roll=1"; do
  ORPHEUS_TUNE="$t" ORPHEUS_BENCH_CTX=2048 ORPHEUS_BENCH_LONG=1 ORPHEUS_BENCH_GREEDY_ONLY=1 timeout 300 python profiles/orpheus_bench.py 2>&1 | grep -E "ms/step" | sed 's/layers=28 prompt 32 + 64 tokens.*-> //' | cut -c1-60 | sed "s/^/[ctx 2048, $t] /" | tee -a $O/orpheus_bench_call22.txt
done; done
for t in "attn_wave=0" "attn_roll=0" "attn_roll=1"; do
  ORPHEUS_TUNE="$t" ORPHEUS_BENCH_GREEDY_ONLY=1 timeout 300 python profiles/orpheus_bench.py 2>&1 | grep -E "ms/step" | sed 's/layers=28 prompt 32 + 64 tokens.*-> //' | cut -c1-60 | sed "s/^/[ctx 1024, $t] /" | tee -a $O/orpheus_bench_call22.txt
done

# round 5, call 5: attn_gqa_wave_kernel (the split decode attention of the Orpheus step as one round trip + one barrier): tests, step time, kernel by kernel
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_orpheus.py tests/test_gpu_gemv_rows.py tests/test_gpu_dia.py tests/test_gpu_upstream.py -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 | tee $O/orpheus_tests_call5.txt
for t in 1 0; do echo "attn_wave=$t"; ORPHEUS_BENCH_GREEDY_ONLY=1 ORPHEUS_TUNE=attn_wave=$t timeout 200 python profiles/orpheus_bench.py 2>&1 | grep -E "ms/step"; done | tee $O/orpheus_bench_call5.txt
(cd /tmp && export TMPDIR=/tmp && ORPHEUS_BENCH_GREEDY_ONLY=1 TTS_HIP_LLAMA_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_orph -- python $R/profiles/orpheus_bench.py > $O/orpheus_kt.log 2>&1; f=$(find /tmp/kt_orph -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_orpheus_call5.csv)
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5/kernel_stats_orpheus_call5.csv')))
for r in rows[:10]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.2f} us  {r['Percentage']}%")
PY

mkdir -p gpurun_out/r4
O=$GRAFT_REPO_ROOT/gpurun_out/r4
timeout 600 python -m pytest tests/test_gpu_orpheus.py tests/test_gpu_gemv_rows.py tests/test_gpu_dia.py tests/test_gpu_upstream.py -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8
timeout 200 python profiles/orpheus_bench.py 2>&1 | grep -E "ms/step|GB/s"
(cd /tmp && export TMPDIR=/tmp && TTS_HIP_LLAMA_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_orph -- python $GRAFT_REPO_ROOT/profiles/orpheus_bench.py > $O/orpheus_kt2.log 2>&1; f=$(find /tmp/kt_orph -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_orpheus_call7.csv)
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4/kernel_stats_orpheus_call7.csv')))
for r in rows[:12]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.2f} us  {r['Percentage']}%")
PY

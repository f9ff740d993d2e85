# round 5, call 1: (a) the new parity tests of the row-major self-attention at the measured history; (b) Orpheus with its staging inputs requested
# before the weights + the straight-line combine: tests, ms/step, kernel by kernel; (c) pipe counters of gemm_tile_kernel at 1024 rows (NEXT.md item 4)
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parler.py -q -k "row_major or measured_history" 2>&1 | tail -25 > $O/parity_attn_rows.txt; tail -5 $O/parity_attn_rows.txt
timeout 600 python -m pytest tests/test_gpu_orpheus.py tests/test_gpu_gemv_rows.py tests/test_gpu_dia.py -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 | tee $O/orpheus_tests.txt
ORPHEUS_BENCH_GREEDY_ONLY=1 timeout 200 python profiles/orpheus_bench.py 2>&1 | grep -E "ms/step|GB/s" | tee $O/orpheus_bench_call1.txt
(cd /tmp && export TMPDIR=/tmp && ORPHEUS_BENCH_GREEDY_ONLY=1 TTS_HIP_LLAMA_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_orph -- python $R/profiles/orpheus_bench.py > $O/orpheus_kt.log 2>&1; f=$(find /tmp/kt_orph -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_orpheus_call1.csv)
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5/kernel_stats_orpheus_call1.csv')))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.2f} us  {r['Percentage']}%")
PY
# pipe counters: one pass per group (the guide: counters in their own runs, with --kernel-trace only)
hipcc --offload-arch=gfx950 -O3 -o /tmp/gemm_bench profiles/gemm_bench.hip 2>/dev/null || cp profiles/gemm_bench /tmp/gemm_bench
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  GEMM_BENCH_NBUF=1 GEMM_BENCH_ONLY=product timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_gemm_$i -- /tmp/gemm_bench 1024 > $O/gemm_pmc_$i.log 2>&1
  f=$(find /tmp/pmc_gemm_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/gemm_pmc_$i.csv
done
cd $R
python profiles/tools/pmc_gemm_summary.py $O/gemm_pmc_1.csv $O/gemm_pmc_2.csv $O/gemm_pmc_3.csv | tee $O/gemm_tile_pipe_counters.txt
rm -f $O/gemm_pmc_*.csv

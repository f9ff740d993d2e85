# round 5, call 16: the Dia step after (1) every slab of rms_fold_rows_kernel in one round trip, (2) llama_rope_kv_kernel's loads ahead of its theta chain,
# (3) gemv_stream_kernel<4, PRO_SILU> instantiated per slab count, with gate|up writing 4 (as before) / 2 / 1 slabs (tune dia_gu_slabs)
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dia.py tests/test_gpu_orpheus.py tests/test_gpu_gemv_rows.py -q -x 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 | tee $O/dia_tests_call16.txt
cd /tmp && export TMPDIR=/tmp
for s in 8 2 1; do
  echo "== dia_gu_slabs $s" | tee -a $O/dia_step_kernels_call16.txt
  rm -rf /tmp/prof_dia
  DIA_TUNE="{\"dia_gu_slabs\": $s}" timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dia -- python $R/profiles/dia_bench.py 64 2>&1 | grep -E "lock-step|error" | tee -a $O/dia_step_kernels_call16.txt
  t=$(find /tmp/prof_dia -name "*kernel_trace.csv" | head -1)
  python $R/profiles/tools/trace_steps.py "$t" dia_embed_kernel 32 | tee -a $O/dia_step_kernels_call16.txt
  DIA_TUNE="{\"dia_gu_slabs\": $s}" timeout 300 python $R/profiles/dia_bench.py 64 2>&1 | grep -E "lock-step" | tee -a $O/dia_step_kernels_call16.txt
done

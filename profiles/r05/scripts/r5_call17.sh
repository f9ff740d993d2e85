# round 5, call 17: gemv_stream_kernel<.., DEEP> (all four chunks of the Dia down projection's item in flight): Dia tests, the step by kernel with and without
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dia.py -q -x 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 | tee $O/dia_tests_call17.txt
cd /tmp && export TMPDIR=/tmp
for s in 1 0 1 0; do
  DIA_TUNE="{\"stream_deep\": $s}" timeout 300 python $R/profiles/dia_bench.py 64 2>&1 | grep -E "lock-step|positions 4" | sed "s/^/stream_deep $s: /" | tee -a $O/dia_step_kernels_call17.txt
done
rm -rf /tmp/prof_dia
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dia -- python $R/profiles/dia_bench.py 64 2>&1 | grep -E "lock-step|error" | tee -a $O/dia_step_kernels_call17.txt
t=$(find /tmp/prof_dia -name "*kernel_trace.csv" | head -1)
python $R/profiles/tools/trace_steps.py "$t" dia_embed_kernel 32 | tee -a $O/dia_step_kernels_call17.txt

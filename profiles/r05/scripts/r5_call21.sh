# round 5, call 21: the Dia step, same box: cross-attention through attn_gqa_split_kernel (attn_wave 0) / attn_gqa_wave_kernel<128, 3, EXT> with plain loads / with non-temporal loads
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dia.py -q -x 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 | tee $O/dia_tests_call21.txt
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for t in '{"attn_wave": 0}' '{"attn_nt": 0}' '{"attn_nt": 1}'; do
  DIA_TUNE="$t" timeout 300 python $R/profiles/dia_bench.py 64 2>&1 | grep -E "lock-step" | sed "s/^/$t: /" | tee -a $O/dia_step_kernels_call21.txt
done; done
for t in '{"attn_wave": 0}' '{"attn_nt": 0}' '{"attn_nt": 1}'; do
  rm -rf /tmp/prof_dia
  DIA_TUNE="$t" timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dia -- python $R/profiles/dia_bench.py 64 > /dev/null 2>&1
  tr=$(find /tmp/prof_dia -name "*kernel_trace.csv" | head -1)
  python $R/profiles/tools/trace_steps.py "$tr" dia_embed_kernel 32 | grep -E "steps|attn_gqa" | sed "s/^/$t: /" | tee -a $O/dia_step_kernels_call21.txt
done

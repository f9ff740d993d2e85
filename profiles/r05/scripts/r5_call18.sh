# round 5, call 18 (second run: key slice in blockIdx.x): attn_gqa_kernel / attn_gqa_split_kernel with the query, the first K batch and the first V batch requested together (the query is folded
# and rotated under them): Dia + Orpheus tests, the Dia step by kernel
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dia.py tests/test_gpu_orpheus.py -q -x 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 | tee $O/dia_tests_call18.txt
cd /tmp && export TMPDIR=/tmp
for s in 1 2; do timeout 300 python $R/profiles/dia_bench.py 64 2>&1 | grep -E "lock-step|positions 4" | tee -a $O/dia_step_kernels_call18.txt; done
rm -rf /tmp/prof_dia
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dia -- python $R/profiles/dia_bench.py 64 2>&1 | grep -E "lock-step|error" | tee -a $O/dia_step_kernels_call18.txt
t=$(find /tmp/prof_dia -name "*kernel_trace.csv" | head -1)
python $R/profiles/tools/trace_steps.py "$t" dia_embed_kernel 32 | tee -a $O/dia_step_kernels_call18.txt

# round 5, call 20 (fourth run: the third with non-temporal loads; third run: eight passes through THREE rolling slots = 112 registers, four workgroups per CU = 28.2 us; second run: four slots, 130 registers = 27.7 us; first: all eight at once = 30.3 us)
# and rotated under them): Dia + Orpheus tests, the Dia step by kernel
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dia.py tests/test_gpu_orpheus.py -q -x 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 | tee $O/dia_tests_call20.txt
cd /tmp && export TMPDIR=/tmp
for s in 1 2; do timeout 300 python $R/profiles/dia_bench.py 64 2>&1 | grep -E "lock-step|positions 4" | tee -a $O/dia_step_kernels_call20.txt; done
rm -rf /tmp/prof_dia
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dia -- python $R/profiles/dia_bench.py 64 2>&1 | grep -E "lock-step|error" | tee -a $O/dia_step_kernels_call20.txt
t=$(find /tmp/prof_dia -name "*kernel_trace.csv" | head -1)
python $R/profiles/tools/trace_steps.py "$t" dia_embed_kernel 32 | tee -a $O/dia_step_kernels_call20.txt

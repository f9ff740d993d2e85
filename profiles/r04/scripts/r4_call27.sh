# round 4, call 27: the folded one-sequence steps (Orpheus: slice merge in the o projection; Dia: slice merge + silu * up in the projections' staging;
# gemv_stream_kernel's unconditional first weight loads) — their tests first, then the whole GPU suite, smoke(), the default bench line
mkdir -p gpurun_out/r4
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4
timeout 400 python -m pytest tests/test_gpu_orpheus.py tests/test_gpu_dia.py -q -x --tb=short 2>&1 | tail -30 > $O/gpu_tests_call27_folds.txt
cat $O/gpu_tests_call27_folds.txt | tail -30
if ! grep -q " passed" $O/gpu_tests_call27_folds.txt || grep -q "failed\|error" $O/gpu_tests_call27_folds.txt; then echo "FOLD TESTS NOT GREEN: stopping"; exit 0; fi
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_orpheus.py --deselect tests/test_gpu_dia.py 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 > $O/gpu_tests_call27_rest.txt; cat $O/gpu_tests_call27_rest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --no-long > $O/bench_default_call27.json 2> $O/bench_default_call27.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4/bench_default_call27.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'])
print('roofline', {k: d['roofline'].get(k) for k in ('bound', 'achieved', 'frac', 'avg_launch_us')})
print('b1', d.get('decode_step_batch1', {}).get('steps_1024'))
print('e2e', json.dumps(d.get('generate_batch1_end_to_end'))[:400])
print('secondary', {k: (v.get('value'), v.get('ms_per_decode_step')) for k, v in d.get('secondary', {}).items()})

PY

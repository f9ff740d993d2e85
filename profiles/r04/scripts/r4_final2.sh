# round 4, last call: the final library (Dia's folding prologues on, Llama's merge back on the combine launch): whole GPU suite, smoke(), the default bench line
mkdir -p gpurun_out/r4
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4
timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 > $O/gpu_tests_final2.txt; cat $O/gpu_tests_final2.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --time-budget-s 0 > $O/bench_default_final2.json 2> $O/bench_default_final2.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4/bench_default_final2.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'])
print('roofline', {k: d['roofline'].get(k) for k in ('bound', 'achieved', 'frac', 'avg_launch_us')})
print('b1', d.get('decode_step_batch1', {}).get('steps_1024'))
print('e2e', {k: v.get('x_real_time') for k, v in d.get('generate_batch1_end_to_end', {}).items() if isinstance(v, dict)})
print('secondary', {k: (v.get('value'), v.get('ms_per_decode_step')) for k, v in d.get('secondary', {}).items()})
print('long', {k: (v.get('audio_seconds_per_sec'), v.get('of_uniform'), v.get('of_lockstep_ragged')) for k, v in d.get('long_utterances', {}).items() if isinstance(v, dict) and 'audio_seconds_per_sec' in v})
PY

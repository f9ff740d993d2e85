# attn_short with rows sharing a wave's K / V, convT bias loads up front: parity + per-class times of one 1024-row runner and the codec
mkdir -p gpurun_out/r3
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
ulimit -c 0
{
timeout 600 python -m pytest tests/test_gpu_parler.py tests/test_gpu_runner.py tests/test_gpu_dac.py tests/test_gpu_upstream.py -q -x 2>&1 | tail -3
timeout 300 python profiles/dac_bench.py 248 3 --batch=64 --prof 2>&1 | tail -7
timeout 300 python bench.py --streams 1 --batch 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-step-sweep --no-long --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('value', d['value'])
kc = d['kernel_classes']; tot = sum(v['ms'] for v in kc.values())
for k, v in sorted(kc.items(), key=lambda kv: -kv[1]['ms']): print(f'  {k:16s} {v[\"ms\"]:8.1f} ms {v[\"launches\"]:6d} {100 * v[\"ms\"] / tot:5.1f}%')
for f in d['roofline_families']: print('  ', f['kernel'][:70], f['bound'], f['achieved'], f['frac'])
"
} > gpurun_out/r3/attn_short_rows_call34.txt 2>&1
cat gpurun_out/r3/attn_short_rows_call34.txt

mkdir -p gpurun_out/r3
TTS_HIP_MAX_ROWS=1152 timeout 400 python bench.py --batch 1152 --streams 1 --steps 1 --warmup 1 --no-step-sweep --no-cpu-baseline --no-long --no-secondary 2>/dev/null > gpurun_out/r3/bench_b1152_s1.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3/bench_b1152_s1.json').read().strip().split('\n')[-1])
print('value', d['value'])
for k, v in d['kernel_classes'].items():
    print('%-16s %9.1f ms %6d launches %8.2f us/launch %8.1f GB/s %8.1f TF' % (k, v['ms'], v['launches'], v['ms'] / v['launches'] * 1e3, v['GBps'], v['TFLOPs']))
PY
for cfg in "768 3" "576 3"; do
  set -- $cfg
  TTS_HIP_MAX_ROWS=1152 timeout 400 python bench.py --batch $1 --streams $2 --steps 1 --warmup 1 --no-roofline --no-step-sweep --no-cpu-baseline --no-long --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('batch $1 streams $2: %.1f audio-s/s  ms_per_step %.0f  ms_per_generate_batch %.0f' % (d['value'], d['ms_per_step'], d['ms_per_generate_batch']))" 2>&1 | tail -1
done

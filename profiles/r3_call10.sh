mkdir -p gpurun_out/r3
{
echo "== DAC + runner tests (shared codec buffers, planes)"
timeout 600 python -m pytest tests/test_gpu_dac.py tests/test_gpu_runner.py tests/test_gpu_snac.py -q -x 2>&1 | tail -3
echo "== many rows with TTS_HIP_MAX_ROWS=1152"
TTS_HIP_MAX_ROWS=1152 timeout 600 python -m pytest tests/test_gpu_parler.py -q -x -k "many_rows" -s 2>&1 | grep -E "rows=|passed|failed|rror" | tail -6
for cfg in "384 3" "768 2" "1152 1" "576 2" "1152 2"; do
  set -- $cfg
  TTS_HIP_MAX_ROWS=1152 timeout 300 python bench.py --batch $1 --streams $2 --steps 2 --warmup 1 --no-roofline --no-step-sweep --no-cpu-baseline --no-long --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('batch $1 streams $2: %.1f audio-s/s  ms_per_step %.0f  ms_per_generate_batch %.0f' % (d['value'], d['ms_per_step'], d['ms_per_generate_batch']))" 2>&1 | tail -1
done
} > gpurun_out/r3/rows_sweep_call10.txt 2>&1
cat gpurun_out/r3/rows_sweep_call10.txt

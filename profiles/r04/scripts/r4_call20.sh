mkdir -p gpurun_out/r4
{
echo "== parity: row-major self-attention (default from 1024 rows; forced from 64 rows with key slices)"
timeout 600 python -m pytest tests/test_gpu_parler.py -q -k "many_rows" 2>&1 | grep -E "passed|failed|^E |worst" | tail -5
TTS_HIP_ATTN_ROWS=64 timeout 600 python -m pytest tests/test_gpu_parler.py -q -k "many_rows or compaction or mid_flight or large_lockstep" 2>&1 | grep -E "passed|failed|^E " | tail -5
for w in 0 1024; do
  echo "== TTS_HIP_ATTN_ROWS=$w"
  TTS_HIP_ATTN_ROWS=$w timeout 300 python profiles/dec_overlap.py 1024 96 2>&1 | head -2
done
for w in 0 1024; do
TTS_HIP_ATTN_ROWS=$w timeout 500 python bench.py --steps 2 --warmup 1 --no-step-sweep --no-cpu-baseline --no-long --no-secondary --no-e2e 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('TTS_HIP_ATTN_ROWS=$w: %.1f audio-s/s  ms_per_step %.0f' % (d['value'], d['ms_per_step']), 'attn', d['roofline']['frac'], d['roofline'].get('avg_launch_us'), d['kernel_classes'].get('attn_self'))"
done
} > gpurun_out/r4/attn_rows.txt 2>&1
cat gpurun_out/r4/attn_rows.txt

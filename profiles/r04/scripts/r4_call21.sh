mkdir -p gpurun_out/r4
for w in 1024 256; do
TTS_HIP_ATTN_ROWS=$w timeout 900 python bench.py --steps 1 --warmup 1 --no-roofline --no-step-sweep --no-cpu-baseline --no-secondary --no-e2e 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
l = d['long_utterances']
print('TTS_HIP_ATTN_ROWS=$w: headline %.1f; long uniform %.1f (attn %s us), ragged %.1f, ragged_stream %.1f' % (d['value'], l['uniform']['audio_seconds_per_sec'], l['uniform'].get('attn_self', {}).get('avg_launch_us'), l['ragged']['audio_seconds_per_sec'], l['ragged_stream']['audio_seconds_per_sec']))"
done > gpurun_out/r4/attn_rows_long.txt 2>&1
cat gpurun_out/r4/attn_rows_long.txt

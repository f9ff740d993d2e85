# row compaction: the parity test, then the long_utterances section (uniform + ragged) with and without it
mkdir -p gpurun_out/r3
{
timeout 300 python -m pytest tests/test_gpu_parler.py -q -x -k "compaction or eos or lockstep or device_resident" 2>&1 | tail -3
for c in 1 0; do
echo "== TTS_HIP_GEN_COMPACT=$c"
TTS_HIP_GEN_COMPACT=$c timeout 900 python bench.py --steps 1 --warmup 1 --no-roofline --no-step-sweep --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('headline', d['value']); print(json.dumps(d['long_utterances']))"
done
} > gpurun_out/r3/compaction_call20.txt 2>&1
cat gpurun_out/r3/compaction_call20.txt

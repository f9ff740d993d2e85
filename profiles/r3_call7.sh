# batch x runner sweep with the round-3 codec (which half bounds the wall now?)
mkdir -p gpurun_out/r3
: > gpurun_out/r3/batch_stream_sweep.txt
for cfg in "384 3" "512 2" "576 2" "512 3" "384 4" "768 1" "384 2"; do
  set -- $cfg
  timeout 300 python bench.py --batch $1 --streams $2 --steps 2 --warmup 1 --no-roofline --no-step-sweep --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('batch $1 streams $2: %.1f audio-s/s  ms_per_step %.0f  ms_per_generate_batch %.0f' % (d['value'], d['ms_per_step'], d['ms_per_generate_batch']))" >> gpurun_out/r3/batch_stream_sweep.txt 2>&1
done
cat gpurun_out/r3/batch_stream_sweep.txt

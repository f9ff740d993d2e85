mkdir -p gpurun_out/r4
run() { label=$1; shift
  env "$@" timeout 500 python bench.py --steps 2 --warmup 1 --no-roofline --no-step-sweep --no-cpu-baseline --no-long --no-secondary --no-e2e $BARGS 2>gpurun_out/r4/rows_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$label: %.1f audio-s/s  ms_per_step %.0f  ms_per_generate_batch %.0f' % (d['value'], d['ms_per_step'], d['ms_per_generate_batch']))" || tail -3 gpurun_out/r4/rows_err.log; }
{
BARGS="--batch 2048 --streams 2" run "2x2048" TTS_HIP_MAX_ROWS=2048
BARGS="--batch 1536 --streams 2" run "2x1536" TTS_HIP_MAX_ROWS=2048
BARGS="--batch 1024 --streams 3" run "3x1024 walk4" TTS_HIP_ATTN_WALK=4
BARGS="--batch 1024 --streams 3" run "3x1024" A=1
} > gpurun_out/r4/rows_sweep_call15.txt 2>&1
cat gpurun_out/r4/rows_sweep_call15.txt

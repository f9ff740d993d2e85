# hardware queues: HIP multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; 3 runners x (decoder + codec stream) = 6 streams
mkdir -p gpurun_out/r4
run() { # label, env..., -- bench args
  label=$1; shift
  env "$@" timeout 400 python bench.py --steps 2 --warmup 1 --no-roofline --no-step-sweep --no-cpu-baseline --no-long --no-secondary $BARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$label: %.1f audio-s/s  ms_per_step %.0f  ms_per_generate_batch %.0f' % (d['value'], d['ms_per_step'], d['ms_per_generate_batch']))"
}
{
BARGS="--batch 1024 --streams 3" run "3x1024 default queues" A=1
BARGS="--batch 1024 --streams 3" run "3x1024 GPU_MAX_HW_QUEUES=8" GPU_MAX_HW_QUEUES=8
BARGS="--batch 1024 --streams 3" run "3x1024 GPU_MAX_HW_QUEUES=16" GPU_MAX_HW_QUEUES=16
BARGS="--batch 512 --streams 6" run "6x512 GPU_MAX_HW_QUEUES=16" GPU_MAX_HW_QUEUES=16
BARGS="--batch 512 --streams 6" run "6x512 default queues" A=1
BARGS="--batch 768 --streams 4" run "4x768 GPU_MAX_HW_QUEUES=16" GPU_MAX_HW_QUEUES=16
} > gpurun_out/r4/hw_queues.txt 2>&1
cat gpurun_out/r4/hw_queues.txt

# attn_short with the prompt length as a compile-time bound (default) and the 4-lanes-per-key experiment: parity subset + launch time at 1024 rows
mkdir -p gpurun_out/r3
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
ulimit -c 0
{
for v in 0 1; do
echo "== TTS_HIP_ATTN_SHORT16=$v"
TTS_HIP_ATTN_SHORT16=$v timeout 100 python -m pytest tests/test_gpu_parler.py tests/test_gpu_upstream.py -q -x -k "prefill_and_steps or lockstep_sequences or conditional or many_rows or upstream_parler" 2>&1 | tail -2
TTS_HIP_ATTN_SHORT16=$v timeout 60 python profiles/attn_short_time.py /tmp/lg_$v.npy 2>&1 | tail -4
done
python -c "
import numpy as np
a, b = np.load('/tmp/lg_0.npy'), np.load('/tmp/lg_1.npy')
print('logits of the two kernels: max |diff|', float(np.abs(a - b).max()), 'max |logit|', float(np.abs(a).max()))"
} > gpurun_out/r3/attn_short16_call37.txt 2>&1
cat gpurun_out/r3/attn_short16_call37.txt

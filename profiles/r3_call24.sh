# batch-1 chain: fc2 split into 256 workgroups + deferred attention combine: parity, ms/step per variant, timeline
mkdir -p gpurun_out/r3
{
timeout 900 python -m pytest tests/test_gpu_parler.py tests/test_gpu_upstream.py -q -x 2>&1 | tail -5
timeout 900 python profiles/b1_chain.py 2>&1 | grep -v Warning | tail -40
} > gpurun_out/r3/b1_chain_call24.txt 2>&1
cat gpurun_out/r3/b1_chain_call24.txt

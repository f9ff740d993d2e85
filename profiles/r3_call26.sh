# batch-1 chain: keys per pass / key splits of the self-attention on top of call 25
mkdir -p gpurun_out/r3
{
timeout 900 python -m pytest tests/test_gpu_parler.py tests/test_gpu_upstream.py -q -x 2>&1 | tail -3
timeout 1200 python profiles/b1_chain.py 2>&1 | grep -v Warning | tail -40
} > gpurun_out/r3/b1_chain_call26.txt 2>&1
cat gpurun_out/r3/b1_chain_call26.txt

# batch-1 chain: straight-line prologues + DPP/permlane reductions + fc2 slabs + deferred combine
mkdir -p gpurun_out/r3
{
profiles/wave_sum_check
timeout 900 python -m pytest tests/test_gpu_parler.py tests/test_gpu_upstream.py -q -x 2>&1 | tail -5
timeout 900 python profiles/b1_chain.py 2>&1 | grep -v Warning | tail -40
} > gpurun_out/r3/b1_chain_call25.txt 2>&1
cat gpurun_out/r3/b1_chain_call25.txt

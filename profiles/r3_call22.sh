# batch-1 chain: parity of the knobs, then ms/step of every variant + the in-kernel timeline
mkdir -p gpurun_out/r3
{
timeout 600 python -m pytest tests/test_gpu_parler.py -q -x -k "batch1_chain or prefill_and_steps or long_context or device_resident or graph_replay" 2>&1 | tail -5
timeout 900 python profiles/b1_chain.py 2>&1 | tail -60
} > gpurun_out/r3/b1_chain_call22.txt 2>&1
cat gpurun_out/r3/b1_chain_call22.txt

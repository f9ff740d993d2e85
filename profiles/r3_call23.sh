# batch-1 chain: in-kernel timeline of one graph-replayed step + key-split variants of the self-attention
mkdir -p gpurun_out/r3
{
timeout 900 python profiles/b1_chain.py 2>&1 | grep -v Warning | tail -60
} > gpurun_out/r3/b1_chain_call23.txt 2>&1
cat gpurun_out/r3/b1_chain_call23.txt

mkdir -p gpurun_out/r4
{
for w in 0 6 5 4 8; do
  echo "== TTS_HIP_ATTN_WALK=$w"
  TTS_HIP_ATTN_WALK=$w timeout 300 python profiles/dec_overlap.py 1024 96 2>&1 | head -2
done
echo "== parity under TTS_HIP_ATTN_WALK=6"
TTS_HIP_ATTN_WALK=6 timeout 600 python -m pytest tests/test_gpu_parler.py -q -k "many_rows or lockstep or compaction" 2>&1 | grep -E "passed|failed|^E " | tail -3
} > gpurun_out/r4/attn_walk.txt 2>&1
cat gpurun_out/r4/attn_walk.txt

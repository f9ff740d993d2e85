for e in "TTS_HIP_ATTN_FUSED=0" "TTS_HIP_ATTN_FUSED=1 TTS_HIP_ATTN_NSPLIT=4" "TTS_HIP_ATTN_FUSED=1 TTS_HIP_ATTN_NSPLIT=8"; do
echo "== $e"; env $e timeout 300 python profiles/step_sweep.py 2>&1 | grep -E "N=" 
done

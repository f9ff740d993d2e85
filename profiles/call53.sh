for e in "TTS_HIP_ATTN_FUSED=0" "TTS_HIP_ATTN_FUSED=1" "TTS_HIP_ATTN_FUSED=1 TTS_HIP_ATTN_NSPLIT=4" "TTS_HIP_ATTN_FUSED=1 TTS_HIP_ATTN_NSPLIT=16"; do
echo "== $e"; env $e timeout 300 python profiles/b1_prof.py 512 2>&1 | grep -E "N="
done
timeout 900 python -m pytest tests/test_gpu_parler.py tests/test_gpu_runner.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -4

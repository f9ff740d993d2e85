for e in "X=1" "TTS_HIP_GEMV_ROWS=1" "TTS_HIP_ATTN_NSPLIT=4" "TTS_HIP_ATTN_NSPLIT=8" "TTS_HIP_GEMV_ROWS=1 TTS_HIP_ATTN_NSPLIT=4"; do
echo "== $e"; env $e timeout 300 python profiles/b1_prof.py 512 2>&1 | grep -E "N=|eager|attn_self|gemm_fc2|gemm_attn_out|gemm_cross_out"
done

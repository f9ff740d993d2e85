mkdir -p gpurun_out/r4
for u in 8 12 16 4; do
  echo "== rows kernel, $u keys in flight per lane"
  TTS_HIP_ATTN_ROWS_U=$u DEC_OVERLAP_ONE=1 timeout 200 python profiles/dec_overlap.py 1024 128 2>&1 | head -1
done > gpurun_out/r4/attn_rows_u.txt 2>&1
cat gpurun_out/r4/attn_rows_u.txt

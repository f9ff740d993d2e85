timeout 900 python -m pytest tests/test_gpu_orpheus.py tests/test_gpu_gemv_rows.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -3
for e in 0 1; do echo "== TTS_HIP_Q4_RMS=$e"; TTS_HIP_Q4_RMS=$e timeout 300 python profiles/orpheus_bench.py 2>&1 | grep -E "ms/step" | head -1; done

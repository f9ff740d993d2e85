timeout 1200 python -m pytest tests/test_gpu_parler.py tests/test_gpu_runner.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -4
for e in 0 1; do echo "== TTS_HIP_CROSS_FUSED=$e"; TTS_HIP_CROSS_FUSED=$e timeout 300 python profiles/step_sweep.py 2>&1 | grep -E "N=" ; done

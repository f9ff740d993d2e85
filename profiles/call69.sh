for e in 0 1 256; do echo "== TTS_HIP_WEIGHT_PREFETCH=$e"; TTS_HIP_WEIGHT_PREFETCH=$e timeout 300 python profiles/b1_prof.py 512 2>&1 | grep -E "N=" ; done
timeout 1200 python -m pytest tests/test_gpu_parler.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -3

mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_orpheus.py tests/test_gpu_dia.py tests/test_gpu_gemv_rows.py -x -q 2>&1 | grep -E "passed|failed|^E " | tail -4
for e in 0 1; do echo "== TTS_HIP_ATTN_FUSED=$e"; TTS_HIP_ATTN_FUSED=$e DIA_BENCH_UTTERANCES=4 timeout 300 python profiles/dia_bench.py 32 2>&1 | grep -E "lock-step"; TTS_HIP_ATTN_FUSED=$e timeout 300 python profiles/orpheus_bench.py 2>&1 | grep -E "ms/step" | head -2; done
timeout 900 python bench.py --workload dia --steps 2 --warmup 1 > gpurun_out/r2/bench_dia.json 2> gpurun_out/r2/bench_dia.log; python -c "
import json; d=json.load(open('gpurun_out/r2/bench_dia.json')); print(d['value'], d['ms_per_decode_step'], d['roofline']['frac'])"

#!/bin/bash
# First GPU call of the next round: everything that was written after round 1's GPU minutes ran out, in one box acquisition.
#   gpurun --timeout 420 -- 'bash profiles/r02/scripts/round2_first_call.sh'
# Each step has its own timeout and log under gpurun_out/r2/; nothing here changes defaults.
set -u
mkdir -p gpurun_out/r2
cd "$(dirname "$0")/.."
run() { name=$1; shift; t=$1; shift; ( timeout "$t" "$@" > "gpurun_out/r2/$name.log" 2>&1; echo "rc=$?" >> "gpurun_out/r2/$name.log" ); tail -3 "gpurun_out/r2/$name.log"; }
# 1. the paths behind knobs / gates: parity first
TTS_TEST_EXPERIMENTAL=1 run gated_tests 90 python -m pytest tests/test_gpu_orpheus.py tests/test_gpu_kokoro.py tests/test_gpu_gemv_rows.py -q -k "captured or runner_from_file or 4_bit or noise_block"
# 2. Orpheus-3B Q4_0 step: default, streaming GEMV rows, + captured step
run orpheus_default 60 python profiles/orpheus_bench.py
TTS_HIP_GEMV_ROWS=1 run orpheus_gemv 60 python profiles/orpheus_bench.py
TTS_HIP_GEMV_ROWS=1 TTS_HIP_Q4_NATIVE=1 run orpheus_gemv_q4 60 python profiles/orpheus_bench.py
TTS_HIP_GEMV_ROWS=1 TTS_HIP_Q4_NATIVE=1 TTS_HIP_LLAMA_GRAPH=1 run orpheus_gemv_q4_graph 60 python profiles/orpheus_bench.py
# 3. Dia-1.6B fp16 step: default vs streaming GEMV rows
run dia_default 40 python profiles/dia_bench.py
TTS_HIP_GEMV_ROWS=1 run dia_gemv 40 python profiles/dia_bench.py
# 4. Kokoro-82M, first measurement (parity-only kernels)
run kokoro 90 python profiles/kokoro_bench.py
# 5. headline workload: larger lock-step batches per context (never measured: 1 x 256, 2 x 256 with the fp16 cache, 2 x 192)
run parler_1x256 150 python bench.py --batch 256 --streams 1 --no-cpu-baseline
run parler_2x256_kvf16 200 python bench.py --batch 256 --streams 2 --kv f16 --no-cpu-baseline
run parler_2x192 200 python bench.py --batch 192 --streams 2 --no-cpu-baseline

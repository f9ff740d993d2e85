mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_gpu_kokoro.py tests/test_gpu_dac.py tests/test_gpu_snac.py -q -s 2>&1 | grep -E "passed|failed|kokoro-82m|kokoro own|Error|assert|stuck" | tail -6
for i in 1 2 3; do timeout 200 python profiles/kokoro_bench.py 2>&1 | tail -1; done
timeout 300 python bench.py --workload kokoro --steps 3 > gpurun_out/r2/bench_kokoro.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r2/bench_kokoro.json')); print(d['value'], d['by_length'], d['cpu_baseline'])"
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profkok3 -- python $R/profiles/kokoro_bench.py > $R/gpurun_out/r2/prof_kok3.log 2>&1
cd $R; f=$(find /tmp/profkok3 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2/kernel_stats_kokoro_82m_round2_final.csv; head -9 gpurun_out/r2/kernel_stats_kokoro_82m_round2_final.csv | cut -c1-130

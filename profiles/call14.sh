mkdir -p gpurun_out/r2
export TMPDIR=/tmp
R=$PWD
cd /tmp && TTS_HIP_LLAMA_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proforph -- python $R/profiles/orpheus_bench.py > $R/gpurun_out/r2/prof_orph.log 2>&1
cd $R; f=$(find /tmp/proforph -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2/kernel_stats_orpheus_3b_q4_0_gemv_nograph.csv; head -14 gpurun_out/r2/kernel_stats_orpheus_3b_q4_0_gemv_nograph.csv | cut -c1-150; tail -3 gpurun_out/r2/prof_orph.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profkok -- python $R/profiles/kokoro_bench.py > $R/gpurun_out/r2/prof_kok.log 2>&1
cd $R; f=$(find /tmp/profkok -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2/kernel_stats_kokoro_82m.csv; head -24 gpurun_out/r2/kernel_stats_kokoro_82m.csv | cut -c1-150

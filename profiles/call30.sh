mkdir -p gpurun_out/r2
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profkok2 -- python $R/profiles/kokoro_bench.py > $R/gpurun_out/r2/prof_kok2.log 2>&1
cd $R; f=$(find /tmp/profkok2 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2/kernel_stats_kokoro_82m_mfma_convs.csv; head -12 gpurun_out/r2/kernel_stats_kokoro_82m_mfma_convs.csv | cut -c1-140

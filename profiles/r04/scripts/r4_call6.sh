mkdir -p gpurun_out/r4
O=$GRAFT_REPO_ROOT/gpurun_out/r4
(cd /tmp && export TMPDIR=/tmp && TTS_HIP_LLAMA_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_orph -- python $GRAFT_REPO_ROOT/profiles/orpheus_bench.py > $O/orpheus_kt.log 2>&1; f=$(find /tmp/kt_orph -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_orpheus_baseline.csv)
tail -4 $O/orpheus_kt.log
head -16 $O/kernel_stats_orpheus_baseline.csv | cut -c1-170
timeout 200 python profiles/orpheus_bench.py 2>&1 | tail -4

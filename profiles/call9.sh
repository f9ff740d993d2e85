mkdir -p gpurun_out/r2
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profdia -- python $R/profiles/dia_bench.py > $R/gpurun_out/r2/prof_dia.log 2>&1
cd $R; f=$(find /tmp/profdia -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2/kernel_stats_dia_1_6b_lockstep4.csv; head -22 gpurun_out/r2/kernel_stats_dia_1_6b_lockstep4.csv | cut -c1-170
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proforph -- python $R/profiles/orpheus_bench.py > $R/gpurun_out/r2/prof_orph.log 2>&1
cd $R; f=$(find /tmp/proforph -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2/kernel_stats_orpheus_3b_q4_0_defaults.csv; head -16 gpurun_out/r2/kernel_stats_orpheus_3b_q4_0_defaults.csv | cut -c1-170

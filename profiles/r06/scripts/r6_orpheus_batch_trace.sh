#!/bin/bash
# kernel trace of the lock-step Orpheus step (3B shapes, Q4_0 matrices) at PROBE_B utterances; tests first
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
[ -n "$SKIP_TESTS" ] || python -m pytest tests/test_gpu_orpheus.py -x -q 2>&1 | grep -v Warning | tail -4 > $O/orpheus_tests.txt
cd /tmp && export TMPDIR=/tmp
PROBE_B=${PROBE_B:-8} PROBE_TUNE=${PROBE_TUNE:-q_stream=1} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o ob -- python $R/profiles/orpheus_batch_probe.py > $O/orpheus_batch_kt.log 2>&1
python $R/profiles/tools/kstats_top.py $(find /tmp/kt -name '*kernel_stats.csv' | head -1) 24 > $O/orpheus_batch_kernels.txt
cat $O/orpheus_tests.txt; grep ms/step $O/orpheus_batch_kt.log; cat $O/orpheus_batch_kernels.txt

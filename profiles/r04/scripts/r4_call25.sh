mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -6 > $O/gpu_tests_final.txt; cat $O/gpu_tests_final.txt
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt3 -- python $R/bench.py --batch 1024 --streams 3 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-step-sweep --no-long --no-secondary --no-e2e > $O/kt_bench_s3.log 2>&1; f=$(find /tmp/kt3 -name '*kernel_trace.csv' | head -1); python $R/profiles/overlap_timeline.py "$f" > $O/timeline_3runners.txt 2>&1; cat $O/timeline_3runners.txt)

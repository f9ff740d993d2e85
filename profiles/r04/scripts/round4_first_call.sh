# The GPU call round 4 opens with (nothing here has been run): refresh the evidence that the last library of round 3 changed but the
# committed line predates (attn_short_kernel with a compile-time prompt bound), the PMC traffic of the kernels whose epilogues changed, and the
# first measurement for §8 item 4 (Orpheus: the step's launches seen from the inside).
mkdir -p gpurun_out/r4
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
ulimit -c 0
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 > gpurun_out/r4/gpu_tests_first.txt
cat gpurun_out/r4/gpu_tests_first.txt
timeout 1800 python bench.py > gpurun_out/r4/bench_default_first.json 2> gpurun_out/r4/bench_default_first.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4/bench_default_first.json').read().strip().split('\n')[-1])
print('value', d['value'], 'attn', d['roofline']['frac'], 'b1', d['decode_step_batch1']['steps_1024'])
for f in d['roofline_families']: print('  ', f['kernel'][:70], f['bound'], f['achieved'], f['frac'], f['share_of_kernel_time'])
PY
# PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of the codec pass and the 1024-row decoder loop -> gpurun_out/r3/pmc/pmc_traffic.json
bash profiles/r03/scripts/r3_pmc.sh > gpurun_out/r4/pmc_run.txt 2>&1; tail -30 gpurun_out/r4/pmc_run.txt | cut -c1-160
# Orpheus-3B Q4_0 step, kernel by kernel (eager: rocprofv3 cannot follow its graph replays)
(cd /tmp && export TMPDIR=/tmp && TTS_HIP_LLAMA_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_orph -- python $GRAFT_REPO_ROOT/profiles/orpheus_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r4/orpheus_kt.log 2>&1; f=$(find /tmp/kt_orph -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r4/kernel_stats_orpheus_first.csv; head -14 $GRAFT_REPO_ROOT/gpurun_out/r4/kernel_stats_orpheus_first.csv | cut -c1-150)

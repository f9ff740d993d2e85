#!/bin/bash
# free-running runners against runners joined after every batch (the rounds 1-5 loop): 6 timed steps each
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
F="--steps 6 --warmup 1 --no-long --no-secondary --no-e2e --no-step-sweep --no-cpu-baseline --no-roofline"
for m in "--join-steps" ""; do
  timeout 900 python bench.py $F $m > $O/bench_join_${m:+joined}.json 2> $O/bench_join_${m:+joined}.err
  python - "$O/bench_join_${m:+joined}.json" "${m:-free-running}" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "value", j["value"], "ms_per_step", j["ms_per_step"], "ms_per_generate_batch", j["ms_per_generate_batch"])
PY
done

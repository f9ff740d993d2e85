#!/bin/bash
# the headline's lock-step shape: rows per forward x contexts per GPU (same flags otherwise; 2 timed steps)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
F="--steps 2 --warmup 1 --no-long --no-secondary --no-e2e --no-step-sweep --no-cpu-baseline --no-roofline"
run() { # name, env rows, batch, streams
  TTS_HIP_MAX_ROWS=$2 timeout 900 python bench.py --batch $3 --streams $4 $F > $O/bench_shape_$1.json 2> $O/bench_shape_$1.err
  python - "$1" <<'PY'
import json, sys
n = sys.argv[1]
try:
    j = json.loads([l for l in open(f"gpurun_out/r6/bench_shape_{n}.json") if l.startswith("{")][-1])
    print(n, "value", j["value"], "ms_per_step", j["ms_per_step"], "utterances", j["config"].get("utterances_per_gpu"))
except Exception as e:
    print(n, "failed:", e, open(f"gpurun_out/r6/bench_shape_{n}.err").read()[-600:])
PY
}
for spec in "$@"; do IFS=: read name rows batch streams <<< "$spec"; run $name $rows $batch $streams; done

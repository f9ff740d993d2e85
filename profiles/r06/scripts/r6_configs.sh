# round 6: the other configurations of the same line with the kernels of the end of the round (2 timed steps each, no extra sections):
# top-k 50 sampling on the device, Q5_0 and Q4_0 decoder weights, the fp16 KV cache, the F16 codec
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
X="--steps 2 --warmup 1 --no-long --no-secondary --no-e2e --no-step-sweep --no-cpu-baseline"
for cfg in "--sample" "--wtype q5_0" "--wtype q4_0" "--wtype q8_0" "--kv f16" "--dac-wtype f16" "--wtype q5_0 --dac-wtype f16"; do
  n=$(echo $cfg | tr -d ' -' )
  timeout 300 python bench.py $X $cfg > $O/bench_cfg_$n.json 2> $O/bench_cfg_$n.err; echo "[$cfg] rc=$?"
  python - "$O/bench_cfg_$n.json" "$cfg" <<'PY'
import json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith('{')]
if lines:
    d = json.loads(lines[-1])
    r = d.get("roofline") or {}
    print(f"[{sys.argv[2]}] {d['value']} {d['unit']}  ms_per_step {d['ms_per_step']}  dtype {d['dtype']}  roofline {r.get('kernel','')[:50]} frac {r.get('frac')}")
PY
done 2>&1 | tee $O/bench_other_configs.txt

# round 5, call 3: gate|up with both items' weights requested before the prologue (Orpheus tests + step); the whole bench line without a time budget
# (3 timed steps): what every section costs now (time_budget.sections) and the long_utterances parts in their new order
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_orpheus.py tests/test_gpu_gemv_rows.py -q 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -5
ORPHEUS_BENCH_GREEDY_ONLY=1 timeout 200 python profiles/orpheus_bench.py 2>&1 | grep -E "ms/step" | tee $O/orpheus_bench_call3.txt
timeout 1500 python bench.py --time-budget-s 0 > $O/bench_full_call3.json 2> $O/bench_full_call3.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5/bench_full_call3.json') if l.startswith('{')][-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"])
print("roofline", {k: d["roofline"][k] for k in ("kernel","achieved","frac","traffic") if k in d["roofline"]})
print("time_budget", d["time_budget"])
lu=d.get("long_utterances",{})
for k in ("uniform","ragged_stream","uniform_same_mix","ragged"):
    print(k, {kk: vv for kk, vv in lu.get(k,{}).items() if kk not in ("note","workload_note")})
print("b1", d.get("decode_step_batch1",{}).get("steps_1024"))
print("e2e", d.get("generate_batch1_end_to_end",{}).get("top_k_50"))
for n,v in d.get("secondary",{}).items(): print(n, v.get("value"), v.get("unit"), v.get("ms_per_decode_step"), v.get("error"))
PY

# round 5, last call: the whole GPU suite once more (the Dia wave / split test with its corrected bar), smoke(), and the full lines of the secondary
# workloads (python bench.py --workload dia|orpheus|kokoro: their own roofline and cpu_baseline objects)
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -rP 2>&1 | grep -E "passed|failed|^E |^FAILED|rror|streams identical|relative logit error by cached" | tail -20 | tee $O/gpu_tests_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke_final.txt
for w in dia orpheus kokoro; do
  timeout 400 python bench.py --workload $w > $O/bench_workload_$w.json 2> $O/bench_workload_$w.err; echo "$w rc=$?"
  python - $w <<'PY'
import json, sys
w = sys.argv[1]
lines = [l for l in open(f'gpurun_out/r5/bench_workload_{w}.json') if l.startswith('{')]
if lines:
    d = json.loads(lines[-1])
    print(w, d.get("metric"), d.get("value"), d.get("unit"), "ms_per_step", d.get("ms_per_step"), "roofline", {k: d.get("roofline", {}).get(k) for k in ("bound", "achieved", "frac")}, "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
done

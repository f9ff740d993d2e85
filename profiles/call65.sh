mkdir -p gpurun_out/r2
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E |^FAILED|error" | tail -4
for w in dia orpheus kokoro; do timeout 600 python bench.py --workload $w --steps 2 --warmup 1 > gpurun_out/r2/bench_$w.json 2> gpurun_out/r2/bench_$w.log; echo "$w rc=$?"; done
python - <<'PY'
import json
for w in ("dia","orpheus","kokoro"):
    d=json.load(open(f'gpurun_out/r2/bench_{w}.json'))
    print(w, d['value'], d.get('ms_per_decode_step'), (d.get('roofline') or {}).get('frac'))
PY

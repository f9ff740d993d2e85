mkdir -p gpurun_out/r2
S=$(date +%s)
timeout 98 python bench.py --no-step-sweep > gpurun_out/r2/bench_default_end_of_round.json 2> gpurun_out/r2/bench_default_end_of_round.log
echo "rc $? elapsed $(( $(date +%s) - S )) s"
python -c "
import json; d=json.load(open('gpurun_out/r2/bench_default_end_of_round.json')); print(d['value'], d['roofline'], d['cpu_baseline'])"

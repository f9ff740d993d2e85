mkdir -p gpurun_out/r4
bash profiles/r04/scripts/r4_pmc.sh > gpurun_out/r4/pmc_main_run.txt 2>&1; tail -25 gpurun_out/r4/pmc_main_run.txt | cut -c1-170
python - <<'PY'
import json
d = json.load(open('gpurun_out/r4/pmc_main/pmc_traffic.json'))
for k, v in d['kernels'].items(): print(k, round(v['hbm_bytes_per_launch'] / 1e6, 1), 'MB', v['launches'])
PY

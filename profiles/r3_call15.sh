# batch 1: where a step's 1.25 ms go — per-kernel durations of the eager step (rocprofv3 kernel trace) and the event-timed classes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
B1_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/b1 -- python $R/profiles/b1_prof.py 512 > $O/b1_prof_eager.log 2>&1
f=$(find /tmp/b1 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_b1_eager.csv
f=$(find /tmp/b1 -name "*kernel_trace.csv" | head -1); python - "$f" > $O/b1_gaps.txt <<'PY'
import csv, sys
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r[0])
# the last 64 eager steps: take the last 64*196 kernels
tail = rows[-64 * 196:]
busy = sum(e - s for s, e, _ in tail)
span = tail[-1][1] - tail[0][0]
gaps = [tail[i + 1][0] - tail[i][1] for i in range(len(tail) - 1)]
print(f"last {len(tail)} launches: span {span / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us ({busy / span * 100:.1f} %), mean gap {sum(gaps) / len(gaps) / 1e3:.2f} us")
from collections import defaultdict
d = defaultdict(lambda: [0, 0])
for s, e, n in tail:
    d[n[:70]][0] += e - s; d[n[:70]][1] += 1
for n, (t, c) in sorted(d.items(), key=lambda kv: -kv[1][0]):
    print(f"{t / c / 1e3:8.2f} us x {c / 64:6.1f}/step  {t / busy * 100:5.1f} %  {n}")
PY
cd $R
tail -25 $O/b1_prof_eager.log
cat $O/b1_gaps.txt

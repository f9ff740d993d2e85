# round 5, call 12: gate|up of the Orpheus step with 256 / 384 / 512 / 1024 workgroups (the rms staging prologue is repeated per workgroup)
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
for g in 512 256 384 1024 512 256; do echo "q4_gu_wgs=$g"; ORPHEUS_BENCH_GREEDY_ONLY=1 ORPHEUS_TUNE=q4_gu_wgs=$g timeout 200 python profiles/orpheus_bench.py 2>&1 | grep -E "ms/step"; done | tee $O/orpheus_gu_grid_call12.txt

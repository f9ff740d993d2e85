# round 5, call 10: the fold at the long workload's 320 rows (the cost model picks 32 x 32 tiles there): forcing the 64 x 64 tile for the cross-q GEMM
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
for m in 1073741824 256 1073741824 256; do timeout 300 python profiles/dec_loop.py 320 1024 cross_fold_min_rows=$m 2>&1 | tail -1; done | tee $O/cross_fold_320_call10.txt
for m in 1073741824 256; do timeout 300 python profiles/dec_loop.py 512 256 cross_fold_min_rows=$m 2>&1 | tail -1; done | tee -a $O/cross_fold_320_call10.txt

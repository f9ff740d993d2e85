# round 5, call 9: cross-attention inside the cross-q GEMM's 64 x 64 tiles (EPI_CROSS): parity, then the 1024-row decoder loop with and without it
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parler.py -q -k "cross_attention_inside or many_rows or tiled_gemm" 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 | tee $O/cross_fold_tests_call9.txt
for f in 1 0 1 0; do timeout 300 python profiles/dec_loop.py 1024 256 cross_fold=$f 2>&1 | tail -1; done | tee $O/cross_fold_loop_call9.txt

# round 5, call 13: the batch-1 chain with the cross-attention inside its out projection's prologue (PRO_CROSS): Parler tests, the chain with stamps, end to end
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parler.py tests/test_gpu_runner.py -q -x -k "not history and not many_rows and not tiled and not inside_the_q" 2>&1 | grep -E "passed|failed|^E |^FAILED|rror" | tail -8 | tee $O/parler_tests_call13.txt
for f in 1 0 1 0; do B1_TUNE="{\"cross_fold\": $f}" B1_LOGITS=/tmp/l.npy B1_TOKENS=/tmp/t.npy timeout 300 python profiles/b1_chain.py --one 2>&1 | tail -1 | cut -c1-120; done | tee $O/b1_cross_fold_call13.txt
B1_ONLY_DEFAULT=1 timeout 600 python profiles/b1_chain.py 2>&1 | tail -12 | tee -a $O/b1_cross_fold_call13.txt

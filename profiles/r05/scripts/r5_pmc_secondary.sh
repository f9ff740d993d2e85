# round 5: the counter passes of profiles/r04/scripts/r4_pmc_secondary.sh with the kernels of the end of round 5
# PMC traffic of the secondary configs (all three carried traffic: null): Orpheus-3B Q4_0 step, Dia-1.6B 4 x 2-row step, Kokoro's MFMA conv family
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5/pmc_secondary; mkdir -p $O
export PMC_JSON_OUT=$O/pmc_traffic_secondary.json; rm -f $PMC_JSON_OUT
for ctr in FETCH_SIZE WRITE_SIZE; do
  ORPHEUS_BENCH_GREEDY_ONLY=1 TTS_HIP_LLAMA_GRAPH=0 timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_or_$ctr -- python $R/profiles/orpheus_bench.py > $O/or_$ctr.log 2>&1
  cp "$(find /tmp/pmc_or_$ctr -name '*counter_collection.csv' | head -1)" $O/or_$ctr.csv
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_dia_$ctr -- python $R/profiles/dia_bench.py 64 > $O/dia_$ctr.log 2>&1
  cp "$(find /tmp/pmc_dia_$ctr -name '*counter_collection.csv' | head -1)" $O/dia_$ctr.csv
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_kk_$ctr -- python $R/profiles/kokoro_bench.py > $O/kk_$ctr.log 2>&1
  cp "$(find /tmp/pmc_kk_$ctr -name '*counter_collection.csv' | head -1)" $O/kk_$ctr.csv
done
cd $R
PMC_SKIP_STEPS=20 python profiles/pmc_step.py orpheus_3b_q4_0_step t5_embed_kernel $O/or_FETCH_SIZE.csv $O/or_WRITE_SIZE.csv
PMC_SKIP_STEPS=12 python profiles/pmc_step.py dia_1_6b_lockstep4_step dia_embed_kernel $O/dia_FETCH_SIZE.csv $O/dia_WRITE_SIZE.csv
python profiles/pmc_step.py kokoro_conv_mfma kk_ $O/kk_FETCH_SIZE.csv $O/kk_WRITE_SIZE.csv conv1d_mfma_kernel
rm -f $O/*.csv
tail -3 $O/or_FETCH_SIZE.log $O/dia_FETCH_SIZE.log | cut -c1-200

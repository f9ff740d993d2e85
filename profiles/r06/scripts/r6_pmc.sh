# round 6: the same counter passes as profiles/r05/scripts/r5_pmc.sh with the kernels of the end of round 6 (fp16 hi + lo codec: two planes instead of three)
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the bench's kernels: one 64-utterance codec pass and the 1024-row decoder loop,
# launch by launch -> profiles/pmc_traffic.json (roofline.traffic of bench.py), plus the kernel-trace stats of the default bench command
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6/pmc_main; mkdir -p $O
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_dac_$ctr -- python $R/profiles/dac_bench.py 248 1 --batch=64 --no-warmup > $O/dac_$ctr.log 2>&1
  f=$(find /tmp/pmc_dac_$ctr -name "*counter_collection.csv" | head -1); cp "$f" $O/dac_$ctr.csv
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_dec_$ctr -- python $R/profiles/pmc_decoder.py 1024 256 > $O/dec_$ctr.log 2>&1
  f=$(find /tmp/pmc_dec_$ctr -name "*counter_collection.csv" | head -1); cp "$f" $O/dec_$ctr.csv
done
cd $R
PMC_JSON_OUT=$O/pmc_traffic.json PMC_WORKLOAD='{"batch": 1024, "audio_steps": 256, "dac_group": 64, "frames": 248, "codec_arith": 7, "command": "profiles/dac_bench.py 248 1 --batch=64 --no-warmup (one 64-utterance codec pass of the bench) + profiles/pmc_decoder.py 1024 256 (the decoder loop of one 1024-utterance runner, launch by launch), round 6 (fp16 hi + lo codec planes, once-built unit operand)"}' \
  python profiles/pmc_summary.py $O/dac_FETCH_SIZE.csv $O/dac_WRITE_SIZE.csv $O/dec_FETCH_SIZE.csv $O/dec_WRITE_SIZE.csv > $O/pmc_fetch_write_round6.txt
head -40 $O/pmc_fetch_write_round6.txt | cut -c1-150
rm -f $O/*.csv
tail -2 $O/dec_FETCH_SIZE.log

# round 6: pipe counters of the codec kernels with the fp16 hi + lo split (one pass per counter group)
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_d_$i -- python $R/profiles/dac_bench.py 248 1 --batch=64 --no-warmup $1 > $O/dac_pmc_$i.log 2>&1
  f=$(find /tmp/pmc_d_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/dac_pmc_$i.csv
done
cd $R
python profiles/tools/pmc_kernel_summary.py "resunit_t7_kernel|conv_b3p_kernel|convt_b3_kernel" $O/dac_pmc_1.csv $O/dac_pmc_2.csv $O/dac_pmc_3.csv $O/dac_pmc_4.csv | tee $O/dac_pipe_counters_$2.txt
rm -f $O/dac_pmc_*.csv

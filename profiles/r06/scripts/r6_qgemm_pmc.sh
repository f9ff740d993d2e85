# round 6: pipe counters of qgemm_tile_kernel at 1024 rows (one pass per counter group; counters in their own runs with --kernel-trace only)
# usage: bash profiles/r06/scripts/r6_qgemm_pmc.sh "<config names for QGEMM_BENCH_ONLY>" <tag>
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1; ulimit -c 0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
ONLY="$1"; TAG="$2"
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  GEMM_BENCH_NBUF=1 QGEMM_BENCH_ONLY="$ONLY" QGEMM_BENCH_SHAPE=fc1 timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_q_$i -- $R/profiles/qgemm_bench 1024 > $O/qgemm_pmc_$i.log 2>&1
  f=$(find /tmp/pmc_q_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/qgemm_pmc_$i.csv
done
cd $R
python profiles/tools/pmc_gemm_summary.py $O/qgemm_pmc_1.csv $O/qgemm_pmc_2.csv $O/qgemm_pmc_3.csv $O/qgemm_pmc_4.csv $O/qgemm_pmc_5.csv | tee $O/qgemm_tile_pipe_counters_$TAG.txt
rm -f $O/qgemm_pmc_*.csv

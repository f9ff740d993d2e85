# per-kernel times and SQ counters of a 64-utterance codec pass with the fused residual units
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/profiles/dac_bench.py 248 1 --batch=64 > $O/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_dac_pass_b64_fused.csv
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1 -- python $R/profiles/dac_bench.py 248 1 --batch=64 --no-warmup > $O/pmc1.log 2>&1
f=$(find /tmp/pmc1 -name "*counter_collection.csv" | head -1); python $R/profiles/pmc_summary.py "$f" > $O/pmc_busy_dac_b64_fused.txt
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc2 -- python $R/profiles/dac_bench.py 248 1 --batch=64 --no-warmup > $O/pmc2.log 2>&1
f=$(find /tmp/pmc2 -name "*counter_collection.csv" | head -1); python $R/profiles/pmc_summary.py "$f" > $O/pmc_lds_dac_b64_fused.txt
cut -c1-200 $O/kernel_stats_dac_pass_b64_fused.csv | head -16
grep -E "resunit|conv1d_mfma_kernel<7" $O/pmc_busy_dac_b64_fused.txt | cut -c1-150
grep -E "resunit" $O/pmc_lds_dac_b64_fused.txt | cut -c1-150
tail -3 $O/pmc2.log

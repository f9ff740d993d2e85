# phase-2 interleave of resunit_t7 (MI=3): parity, then the codec class table
mkdir -p gpurun_out/r3
{
B3_KNOBS="2" timeout 300 python profiles/b3_check.py 2>&1 | tail -12
timeout 600 python -m pytest tests/test_gpu_dac.py tests/test_gpu_upstream.py -q -x 2>&1 | tail -3
timeout 300 python profiles/dac_bench.py 248 3 --batch=64 --prof 2>&1 | tail -30
} > gpurun_out/r3/resunit_phase2_call21.txt 2>&1
cat gpurun_out/r3/resunit_phase2_call21.txt

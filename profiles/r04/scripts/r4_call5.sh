mkdir -p gpurun_out/r4
timeout 500 python profiles/dec_overlap.py 1024 96 > gpurun_out/r4/dec_overlap.txt 2>&1
tail -12 gpurun_out/r4/dec_overlap.txt

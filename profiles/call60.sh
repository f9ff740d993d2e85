# A/B on one box: the shipped library (conv7 kernels bounded to 4 waves per SIMD) against a build with the bound off
cp tts.cpp_amd/libtts_hip.so /tmp/lib_w4.so
for rep in 1 2; do
for v in w4 w1; do
  if [ $v = w1 ]; then cp profiles/_ab/libtts_hip_w1.so tts.cpp_amd/libtts_hip.so; else cp /tmp/lib_w4.so tts.cpp_amd/libtts_hip.so; fi
  echo "== $v"; timeout 300 python profiles/dac_bench.py 248 3 --batch=64 --prof 2>&1 | grep -E "batch=|conv7"
done; done
cp /tmp/lib_w4.so tts.cpp_amd/libtts_hip.so

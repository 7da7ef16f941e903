# The GPU suite aborted twice in eight runs with glibc's "double free or corruption (!prev)" inside the one-object motcpp_bench_threads calls of
# test_f3_checksums[deepocsort]. This runs the files up to that test with the HOST library built under AddressSanitizer (motcpp_amd/lib_asan/, built by
#   g++ -O1 -g -fsanitize=address -fno-omit-frame-pointer ... host/*.cpp) until the report appears.
set -u
mkdir -p gpurun_out
cd /root/repo
export MOTCPP_LIB_DIR=/root/repo/motcpp_amd/lib_asan LD_PRELOAD="/usr/lib/x86_64-linux-gnu/libasan.so.6 /usr/lib/x86_64-linux-gnu/libstdc++.so.6"
export LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib  # (the interposed dlopen loses the caller's RPATH: torch could not find libcaffe2_nvrtc.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:protect_shadow_gap=0:detect_odr_violation=0:log_path=/root/repo/gpurun_out/asan
for i in $(seq 1 ${1:-10}); do
  timeout 900 python -X faulthandler -m pytest tests/test_multirank_cpu.py tests/test_gpu_boosttrack.py tests/test_gpu_botsort_device.py tests/test_gpu_device_lifecycle.py tests/test_gpu_error_isolation.py -m gpu -x -q -s -k "not pooled_object_overflows" > gpurun_out/repro_$i.txt 2>&1
  rc=$?
  echo "iter $i rc $rc $(grep -a 'passed\|failed' gpurun_out/repro_$i.txt | tail -1 | cut -c1-100)"
  if [ $rc -ne 0 ] || ls gpurun_out/asan.* > /dev/null 2>&1; then
    python -c "import sys; print(open(sys.argv[1],errors='replace').read()[:200])" gpurun_out/repro_$i.txt
    grep -av "^  File\|^Extension" gpurun_out/repro_$i.txt | head -c 3000
    for f in gpurun_out/asan.*; do head -c 12000 $f; done
    break
  fi
  rm -f gpurun_out/repro_$i.txt
done

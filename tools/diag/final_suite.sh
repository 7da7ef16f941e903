cd /root/repo; mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r06d_gputests_$i.txt 2>&1
  echo "run $i rc $? $(grep -a 'passed\|failed' gpurun_out/r06d_gputests_$i.txt | tail -1)"
done
for i in 1 2 3; do timeout 300 python tools/diag/stress_device_cycles.py 3000 1 1 2>&1 | grep -v "^Extension\|amdgpu.ids\|^  File" | tail -3; done
python tools/bench_f3.py 64 > gpurun_out/r06d_f3_streams_and_threads.json 2>/dev/null; head -c 600 gpurun_out/r06d_f3_streams_and_threads.json

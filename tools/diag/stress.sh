cd /root/repo
for v in 0 1; do
  bad=0
  for i in 1 2 3 4 5 6 7; do
    timeout 600 python tools/diag/stress_device_cycles.py 3000 0 1 deepocsort $v > gpurun_out/stress_$i.txt 2>&1
    if ! grep -q "no abort" gpurun_out/stress_$i.txt; then bad=$((bad+1)); grep -v "^Extension\|amdgpu.ids" gpurun_out/stress_$i.txt | head -3; fi
  done
  echo "device kept $v: $bad aborts of 7 runs"
done

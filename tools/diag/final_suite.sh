# the driver's GPU tier, several times over: python -m pytest tests -x -q -m gpu
cd /root/repo; mkdir -p gpurun_out
for i in $(seq 1 ${1:-4}); do
  timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/final_gputests_$i.txt 2>&1
  echo "suite run $i rc $? $(grep -a 'passed\|failed' gpurun_out/final_gputests_$i.txt | tail -1)"
done

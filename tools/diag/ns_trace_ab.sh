cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
Q="--no-cpu-baseline --long-run-steps 0 --sweep-streams= --host-input-steps 0 --isolated-steps 0 --parity-streams 0"
for it in 1 2; do
  rm -rf /tmp/kt_ab; ( cd /tmp && MOT_KF_BLK_ITEMS=$it timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_ab -- python /root/repo/bench.py --steps 20 --warmup 5 $Q > /root/repo/gpurun_out/ab_${it}_line.json 2> /dev/null )
  echo "items $it value $(python -c "import json;print(round(json.loads(open('gpurun_out/ab_${it}_line.json').read().strip().splitlines()[-1])['value']))")"
  python tools/rocpd_top_kernels.py /tmp/kt_ab gpurun_out/ab_${it}_kernels.csv 2>&1 | head -12 | cut -c1-130
done

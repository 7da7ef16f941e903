cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_device_lifecycle.py tests/test_gpu_pooled.py tests/test_gpu_primitives.py tests/test_gpu_trackers.py -m gpu -x -q > gpurun_out/ns3_tests.txt 2>&1; echo "tests: $(grep -a "passed\|failed" gpurun_out/ns3_tests.txt | tail -1)"
for it in 1 2 4 1 2 4; do
MOT_KF_BLK_ITEMS=$it timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-run-steps 0 --sweep-streams= --host-input-steps 0 --isolated-steps 6 --parity-streams 8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); ki=j['kernels_isolated']
print('items $it NS', round(j['value']), 'isolated:', {a:ki[a].get('avg_launch_ms') for a in ki if isinstance(ki[a],dict)}, 'parity', (j.get('parity') or {}).get('mismatching_stream_frames'), '/', (j.get('parity') or {}).get('stream_frames_checked'))"
done

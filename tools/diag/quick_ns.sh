# quick look on the GPU box: ByteTrack device-lifecycle tests, then the kernel table of one NS sub-batch (6144 streams)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-q}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.txt 2>&1; echo "tests: $(grep -a "passed\|failed" gpurun_out/${TAG}_tests.txt | tail -1)"
Q="--no-cpu-baseline --long-run-steps 0 --sweep-streams= --host-input-steps 0 --isolated-steps 0 --parity-streams 8 --no-outputs-resident"
rm -rf /tmp/kt_q; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_q -- python /root/repo/bench.py --steps 12 --warmup 4 --streams 6144 --pipeline 1 $Q > /root/repo/gpurun_out/${TAG}_line.json 2> /root/repo/gpurun_out/${TAG}_err.txt )
python tools/rocpd_top_kernels.py /tmp/kt_q gpurun_out/${TAG}_kernels.csv 2>&1 | head -16 | cut -c1-150
python -c "
import json,sys
j=json.loads(open('gpurun_out/${TAG}_line.json').read().strip().splitlines()[-1]); print('value', round(j['value']), 'parity', j.get('parity'))"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-run-steps 0 --sweep-streams= --host-input-steps 0 --isolated-steps 4 --parity-streams 16 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NS default', round(j['value']), j['kernels_isolated'], 'parity', (j.get('parity') or {}).get('mismatching_stream_frames'), '/', (j.get('parity') or {}).get('stream_frames_checked'))"

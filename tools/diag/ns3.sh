cd /root/repo; mkdir -p gpurun_out
for it in 1024 640 512 1024 640 512; do
MOT_BT_DUPS_ITEMS=$it timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-run-steps 0 --sweep-streams= --host-input-steps 0 --isolated-steps 6 --parity-streams 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); ki=j['kernels_isolated']
print('items $it NS', round(j['value']), 'isolated:', {a:ki[a].get('avg_launch_ms') for a in ki if isinstance(ki[a],dict)})"
done

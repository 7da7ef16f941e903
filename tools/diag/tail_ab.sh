cd /root/repo
timeout 600 python bench.py --no-cpu-baseline --long-run-steps 0 --host-input-steps 0 --isolated-steps 0 --parity-streams 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=j['stream_sweep']['basetracker_update']
print(round(j['value']))
for k,v in b.items():
    if isinstance(v,dict): print(k, round(v['frames/s']), round(v['ms_per_update_p50'],2), round(v['ms_per_update_p99'],2), v['assignments_fast_path'], v['assignments_declined_to_the_exact_kernel'])"

# what the driver does at round end, three times over: the GPU suite, smoke(), the default bench line
cd /root/repo; mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/final_gputests_$i.txt 2>&1
  echo "suite run $i rc $? $(grep -a 'passed\|failed' gpurun_out/final_gputests_$i.txt | tail -1)"
done
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err; python -c "
import json; j=json.loads(open('gpurun_out/final_bench_default.json').read().strip().splitlines()[-1]); print('bench', round(j['value']), j['unit'], 'ms/step', round(j['ms_per_step'],3), 'roofline', round(j['roofline']['frac'],4), 'stale', j['roofline']['traffic_source']['stale'], 'cpu', round(j['cpu_baseline']['value'],1), 'parity', j['parity']['mismatching_stream_frames'], '/', j['parity']['stream_frames_checked'])"

mkdir -p gpurun_out
python gpurun_lapprof.py
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run() { # env workload S thr pipe steps warm
  env $1 timeout 300 python bench.py --workload $2 --steps $6 --warmup $7 --streams $3 --threads $4 --pipeline $5 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err; python -c "
import json;d=json.load(open('gpurun_out/b.json'));print('$1 $2 S=$3 thr=$4 pipe=$5 fps',round(d['value'],1),'ms/step',round(d['ms_per_step'],2),'busy',round(d['gpu_busy_frac'],3),'host',{k:round(v,2) for k,v in d['host_ms_per_step'].items()},{k:round(v['ms_total']/v['launches'],3) for k,v in d['kernels'].items()})" || tail -5 gpurun_out/b.err
}
run X=1 C2 8192 32 2 20 40
run X=1 NS 8192 64 4 6 35
run X=1 NS 4096 32 2 6 35

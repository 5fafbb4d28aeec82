set -x
cd /root/repo
mkdir -p gpurun_out/r6c
WHAMD_DEBUG_TIMING=1 python scripts/gpu_create_timing.py 200000 20 2>&1 | grep -v "^\[whamd timing\]   " | tail -14 | tee gpurun_out/r6c/create_config2.txt
WHAMD_DEBUG_TIMING=1 python scripts/gpu_create_timing.py 200000 20 2>&1 | grep "^\[whamd timing\]   " | tail -17 | tee -a gpurun_out/r6c/create_config2.txt
WHAMD_DEBUG_TIMING=1 python scripts/gpu_create_timing.py 50000 15 2>&1 | grep -v "^\[whamd timing\]   " | tail -8 | tee gpurun_out/r6c/create_config1.txt
WHAMD_DEBUG_TIMING=1 WHAMD_PLAN_THREADS=2 python scripts/gpu_create_timing.py 50000 15 2>&1 | grep -v "^\[whamd timing\]   " | tail -8 | tee gpurun_out/r6c/create_config1_2threads.txt
python scripts/gpu_wide_ab.py 4000 2>&1 | tee gpurun_out/r6c/wide_ab.txt
python bench.py --configs off --pmc off --cpu-baseline-columns 0 --steps 10 --warmup 3 2>gpurun_out/r6c/headline.err | tail -1 | tee gpurun_out/r6c/headline.json
cat gpurun_out/r6c/headline.err | tail -3
for w in config1 config1_x24 config1_x96 blocks24 config3_x8; do python bench.py --workload $w --sub --pmc off --cpu-baseline-columns 0 --steps 5 --warmup 2 2>gpurun_out/r6c/$w.err | tail -1 > gpurun_out/r6c/$w.json; tail -2 gpurun_out/r6c/$w.err; python -c "
import json,sys; d=json.load(open('gpurun_out/r6c/$w.json')); print('$w', 'value', round(d['value']), 'resident', round(d['value_resident']['value']), 'per_rank', {k:(round(v,1) if isinstance(v,float) else v) for k,v in d['per_rank'][0].items()}, 'rate', {k:(round(v,3) if isinstance(v,float) else v) for k,v in (d.get('create_rate') or {}).items() if k!='what'})"; done

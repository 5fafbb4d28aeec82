cd /root/repo
mkdir -p gpurun_out/r6g
WHAMD_DEBUG_TIMING=1 python scripts/gpu_create_timing.py 200000 20 2>&1 | grep -v "^\[whamd timing\]   " | tail -5
WHAMD_DEBUG_TIMING=1 python scripts/gpu_create_timing.py 50000 15 2>&1 | grep -v "^\[whamd timing\]   " | tail -4
python bench.py --configs off --pmc off --cpu-baseline-columns 0 --steps 10 --warmup 3 2>&1 >/dev/null | grep "bench rank"
for w in config1 config1_x24 config1_x96 blocks24 config3 config3_x8; do python bench.py --workload $w --sub --pmc off --cpu-baseline-columns 0 --steps 5 --warmup 2 2>gpurun_out/r6g/$w.err | tail -1 > gpurun_out/r6g/$w.json; grep "bench rank" gpurun_out/r6g/$w.err | sed "s/blocks \[[^]]*\]//"; python -c "
import json,sys; d=json.load(open('gpurun_out/r6g/$w.json')); print('$w', 'value', round(d['value']), 'resident', round(d['value_resident']['value']), 'shape', d['per_rank'][0]['create_threads'], d['per_rank'][0]['host_threads_per_create'], 'tried', [(t['create_threads'], t['host_threads_per_create'], round(t['wall_ms'])) for t in d['host_shapes_tried']], 'rate', {k:(round(v,3) if isinstance(v,float) else v) for k,v in (d.get('create_rate') or {}).items() if k!='what'})"; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4

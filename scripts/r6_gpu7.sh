cd /root/repo
mkdir -p gpurun_out/r6f
python bench.py --configs off --pmc off --cpu-baseline-columns 0 --steps 10 --warmup 3 2>&1 >/dev/null | grep "bench rank"
for w in config1 config1_x24 config1_x96 blocks24 irregular_x24; do python bench.py --workload $w --sub --pmc off --cpu-baseline-columns 0 --steps 5 --warmup 2 2>gpurun_out/r6f/$w.err | tail -1 > gpurun_out/r6f/$w.json; grep "bench rank" gpurun_out/r6f/$w.err | sed "s/blocks \[[^]]*\]//"; python -c "
import json,sys; d=json.load(open('gpurun_out/r6f/$w.json')); print('$w', 'value', round(d['value']), 'resident', round(d['value_resident']['value']), 'shape', d['per_rank'][0]['create_threads'], d['per_rank'][0]['host_threads_per_create'], 'tried', [(t['create_threads'], t['host_threads_per_create'], round(t['wall_ms'])) for t in d['host_shapes_tried']], 'rate', {k:(round(v,3) if isinstance(v,float) else v) for k,v in (d.get('create_rate') or {}).items() if k!='what'})"; done

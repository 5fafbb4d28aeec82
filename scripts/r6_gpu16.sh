cd /root/repo
mkdir -p gpurun_out/r6k
python bench.py --configs off --pmc off --cpu-baseline-columns 0 --steps 10 --warmup 3 2>&1 >/dev/null | grep "bench rank"
for w in config1 config1_x24 config1_x96 config3 blocks24; do python bench.py --workload $w --sub --pmc off --cpu-baseline-columns 0 --steps 5 --warmup 2 2>gpurun_out/r6k/$w.err | tail -1 > gpurun_out/r6k/$w.json; grep "bench rank" gpurun_out/r6k/$w.err | sed "s/blocks \[[^]]*\]//"; python -c "
import json,sys; d=json.load(open('gpurun_out/r6k/$w.json')); print('$w', 'value', round(d['value']), 'resident', round(d['value_resident']['value']), 'tried', [(t['create_threads'], t['host_threads_per_create'], t['tables_per_window'], t['windows_on_device'], round(t['wall_ms'])) for t in d['host_shapes_tried']], 'rate', {k:(round(v,3) if isinstance(v,float) else v) for k,v in (d.get('create_rate') or {}).items() if k!='what'})"; done
python scripts/gpu_concurrent_create.py 96 16 2 | head -3
python scripts/gpu_concurrent_create.py 96 32 2 | head -3
python scripts/gpu_concurrent_create.py 96 64 1 | head -3
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3

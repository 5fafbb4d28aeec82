cd /root/repo
mkdir -p gpurun_out/r6j
for w in config1_x24 config1_x96 irregular_x24; do python bench.py --workload $w --sub --pmc off --cpu-baseline-columns 0 --steps 5 --warmup 2 2>gpurun_out/r6j/$w.err | tail -1 > gpurun_out/r6j/$w.json; grep "bench rank" gpurun_out/r6j/$w.err | sed "s/blocks \[[^]]*\]//"; python -c "
import json,sys; d=json.load(open('gpurun_out/r6j/$w.json')); print('$w', 'value', round(d['value']), 'resident', round(d['value_resident']['value']), 'resident ms', round(d['value_resident']['ms_per_step'],1), 'fwd', round(d['rank0']['forward_ms_per_step'],1), 'tried', [(t['create_threads'], t['host_threads_per_create'], t['tables_per_window'], t['windows_on_device'], round(t['wall_ms'])) for t in d['host_shapes_tried']])"; done

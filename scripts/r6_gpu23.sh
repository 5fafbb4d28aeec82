cd /root/repo
echo "== blocks3 in a 32-CPU slice (what a rank of 8 has): N = 1 shapes"
taskset -c 0-15,128-143 python bench.py --workload blocks3 --sub --pmc off --cpu-baseline-columns 0 --steps 5 --warmup 2 2>&1 >/tmp/o.json | grep "bench rank" | sed "s/blocks \[[^]]*\]//"
python -c "
import json; d=json.load(open('/tmp/o.json')); print('value', round(d['value']), 'resident', round(d['value_resident']['value']), [(t['create_threads'], t['host_threads_per_create'], t['tables_per_window'], t['windows_on_device'], round(t['wall_ms'])) for t in d['host_shapes_tried']])"
echo "== the N > 1 path itself: two ranks, 12 blocks each, on one device"
python bench.py --gpus 2 --oversubscribe --steps 3 --warmup 1 2>&1 >/tmp/o2.json | grep "bench rank" | sed "s/blocks \[[^]]*\]//"
python -c "
import json; d=json.loads(open('/tmp/o2.json').read().strip().splitlines()[-1]); print('value', round(d['value']), 'resident', round(d['value_resident']['value']), d['per_rank'])"

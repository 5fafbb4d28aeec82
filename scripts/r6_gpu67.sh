cd /root/repo
one() { python bench.py --workload $1 --configs off --pmc off --cpu-baseline-columns 0 2> /dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['per_rank'][0]
print('$1 $2', round(d['value']), 'ms', round(d['ms_per_step'],1), 'create', round(r['create_ms'],1), 'solve', round(r['solve_ms'],1), 'shape', r.get('create_threads'), r.get('host_threads_per_create'), r.get('tables_per_window'), r.get('windows_on_device'))"; }
for i in 1 2; do
one config1_x96 "library threads by affinity"
WHAMD_HOST_CPUS=16 one config1_x96 "WHAMD_HOST_CPUS=16"
done
one config1_x24 "library threads by affinity"
WHAMD_HOST_CPUS=16 one config1_x24 "WHAMD_HOST_CPUS=16"
one config2 "library threads by affinity"
WHAMD_HOST_CPUS=16 one config2 "WHAMD_HOST_CPUS=16"

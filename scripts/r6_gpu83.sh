cd /root/repo
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
one() { python bench.py --workload $1 --configs off --pmc off --cpu-baseline-columns 0 2> /dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['per_rank'][0]
print('$1', round(d['value']), 'ms', round(d['ms_per_step'],1), 'create', round(r['create_ms'],1), 'solve', round(r['solve_ms'],1), 'resident', round(d['value_resident']['value']), round(d['value_resident']['ms_per_step'],1))"; }
one config1_x96; one config1_x24; one irregular_x24; one config2

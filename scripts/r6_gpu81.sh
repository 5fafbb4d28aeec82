cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_pedslots.py tests/test_gpu_untrusted_group.py tests/test_gpu_untrusted_quartet.py tests/test_gpu_parity.py tests/test_gpu_group.py -m gpu -x -q 2>&1 | tail -2
one() { python bench.py --workload $1 --configs off --pmc off --cpu-baseline-columns 0 2> /dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['per_rank'][0]
print('$1', round(d['value']), 'ms', round(d['ms_per_step'],1), 'create', round(r['create_ms'],1), 'solve', round(r['solve_ms'],1), 'resident', round(d['value_resident']['value']))"; }
one config3; one config3_x8; one config3_distrust; one quartet; one quartet_distrust

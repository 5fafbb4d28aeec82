cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python scripts/gpu_group_step_pieces.py 96 15 50000 2>&1 | grep "^rep"
timeout 1200 python scripts/gpu_group_soak.py 2>&1 | tail -1
WHAMD_SOAK_BLOCKS=40 timeout 900 python scripts/gpu_soak.py 2>&1 | grep -i mismatch | tail -3
for w in config1_x96 config1_x24; do python bench.py --workload $w --sub --pmc off --cpu-baseline-columns 0 --steps 5 --warmup 2 2>&1 >/tmp/o.json | grep "bench rank" | sed "s/blocks \[[^]]*\]//"; python -c "
import json; d=json.load(open('/tmp/o.json')); print('$w value', round(d['value']), 'resident', round(d['value_resident']['value']))"; done
python bench.py --configs off --pmc off --cpu-baseline-columns 0 --steps 10 --warmup 3 2>&1 >/dev/null | grep "bench rank"

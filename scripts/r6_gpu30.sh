cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k "group or parity or headline" 2>&1 | tail -2
for w in config1_x96 config1_x24 irregular_x24; do python bench.py --workload $w --sub --pmc off --cpu-baseline-columns 0 --steps 6 --warmup 2 2>&1 >/tmp/o.json | grep "bench rank" | sed "s/blocks \[[^]]*\]//"; python -c "
import json; d=json.load(open('/tmp/o.json')); print('$w value', round(d['value']), 'resident', round(d['value_resident']['value']))"; done

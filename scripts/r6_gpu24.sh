cd /root/repo
for opt in "" "--option shared_launches=1"; do echo "== irregular $opt"; python bench.py --workload irregular --sub --pmc off --cpu-baseline-columns 0 --steps 5 --warmup 2 $opt 2>&1 >/tmp/o.json | grep "bench rank" | sed "s/blocks \[[^]]*\]//"; python -c "
import json; d=json.load(open('/tmp/o.json')); print('value', round(d['value']), 'resident', round(d['value_resident']['value']), 'launches', d['rank0']['forward_launches_per_step'], 'us', round(d['roofline']['avg_launch_us'],2))"; done
